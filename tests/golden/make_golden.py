#!/usr/bin/env python3
"""Generates tests/golden/mnist_sign1024x*.json from the reference's own plaintext build.

Runs oracle/_ref/mnist_sign1024x{1,2,3}_ptxt.out (built by oracle/Makefile from the sources where
they lie under /root/reference) inside the corresponding net directory, parses the
"Category k: v" lines that main.cpp:132 prints per image, and stores them next to the first rows of
nets/mnist/mnist_data.csv (pixels, label). Also copies the DATA files the nets read: the packed
trained weights var_prep.dat (format: lib/BinOps_enc.cpp:247-305). Only runs where /root/reference
exists; the committed outputs are what travels.
"""
import json
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    csv_rows = [l.strip() for l in open(os.path.join(REF, "nets/mnist/mnist_data.csv")) if l.strip()]
    images = []
    for line in csv_rows[:100]:
        vals = [int(v) for v in line.split(",") if v != ""]
        images.append({"label": vals[0], "pixels": vals[1:785]})
    json.dump({"source": "first 100 rows of nets/mnist/mnist_data.csv (label, 784 pixels)",
               "labels": [im["label"] for im in images], "pixels": [im["pixels"] for im in images]},
              open(os.path.join(HERE, "mnist_images.json"), "w"), separators=(",", ":"))
    for net in ("sign1024x1", "sign1024x2", "sign1024x3"):
        exe = os.path.join(ROOT, "oracle/_ref/mnist_%s_ptxt.out" % net)
        out = subprocess.run([exe], cwd=os.path.join(REF, "nets/mnist", net), capture_output=True, text=True, check=True).stdout
        cats = [int(m.group(2)) for m in re.finditer(r"Category (\d+): (-?\d+)", out)]
        # only sign1024x1's main.cpp prints the logits (main.cpp:132); the deeper nets print the
        # running count and every 10th prediction (main.cpp:109)
        assert len(cats) in (0, 1000), len(cats)
        logits = [cats[10 * i:10 * i + 10] for i in range(100)] if cats else None
        progress = [[int(g) for g in m.groups()] for m in
                    re.finditer(r"correct:\s+(\d+)\s+image_i:\s+(\d+)\s+Label: (\d+)\s+Prediction: (\d+)", out)]
        acc = re.search(r"Correct: ([0-9.]+)%", out).group(1)
        fixture = {
            "source": "reference plaintext flavour (make ptxt), nets/mnist/%s, NUM_SAMPLES=100" % net,
            "accuracy_percent": float(acc),
            "logits": logits,
            "progress_correct_image_label_prediction": progress,
        }
        json.dump(fixture, open(os.path.join(HERE, "mnist_%s.json" % net), "w"), separators=(",", ":"))
        shutil.copyfile(os.path.join(REF, "nets/mnist", net, "var_prep.dat"), os.path.join(HERE, "mnist_%s_var_prep.dat" % net))
        print(net, "accuracy", acc, "image0", logits[0] if logits else progress[:3])


def relu():
    """ReLU nets: the reference's main.cpp runs ONE image and prints only label/prediction
    (nets/mnist/relu1024x1/main.cpp:25,160), so the logits of the first 100 rows come from oracle/ref_logits_driver.cpp
    (our main() around the reference's unmodified HeBNN, plaintext flavour). ONE OpenMP thread: the reference's
    relu_shift loop shares a scratch value between threads (lib/IntFunc.cpp:953) and is not deterministic otherwise."""
    for net in ("relu1024x1", "relu1024x2", "relu1024x3"):
        exe = os.path.join(ROOT, "oracle/_ref/mnist_%s_logits.out" % net)
        out = subprocess.run([exe, "../mnist_data.csv", "100", "relu"], cwd=os.path.join(REF, "nets/mnist", net), capture_output=True,
                             text=True, check=True, env=dict(os.environ, OMP_NUM_THREADS="1")).stdout
        rows = [[int(t) for t in l.split("logits")[1].split()] for l in out.splitlines() if l.startswith("row")]
        assert len(rows) == 100 and all(len(r) == 10 for r in rows)
        json.dump({"source": "reference plaintext flavour of nets/mnist/%s through oracle/ref_logits_driver.cpp, 100 rows of "
                             "nets/mnist/mnist_data.csv, input v/100 - 1, OMP_NUM_THREADS=1" % net, "logits": rows},
                  open(os.path.join(HERE, "mnist_%s.json" % net), "w"), separators=(",", ":"))
        shutil.copyfile(os.path.join(REF, "nets/mnist", net, "var_prep.dat"), os.path.join(HERE, "mnist_%s_var_prep.dat" % net))
        print(net, "image0", rows[0])


def cifar():
    """CIFAR nets: the reference's main.cpp runs NUM_SAMPLES = 1 image (main.cpp:25) and prints only
    label/prediction, so the fixture holds that one prediction plus the first 20 rows of
    nets/cifar/cifar_data.csv and the packed weights."""
    rows = [l.strip() for l in open(os.path.join(REF, "nets/cifar/cifar_data.csv")) if l.strip()]
    images = []
    for line in rows[:20]:
        vals = [int(v) for v in line.split(",") if v != ""]
        images.append({"label": vals[0], "pixels": vals[1:3073]})
    json.dump({"source": "first 20 rows of nets/cifar/cifar_data.csv (label, 3072 values, HWC order)",
               "labels": [im["label"] for im in images], "pixels": [im["pixels"] for im in images]},
              open(os.path.join(HERE, "cifar_images.json"), "w"), separators=(",", ":"))
    for net in ("binarynet", "binarynet_small"):
        exe = os.path.join(ROOT, "oracle/_ref/cifar_%s_ptxt.out" % net)
        out = subprocess.run([exe], cwd=os.path.join(REF, "nets/cifar", net), capture_output=True, text=True, check=True).stdout
        progress = [[int(g) for g in m.groups()] for m in
                    re.finditer(r"correct:\s+(\d+)\s+image_i:\s+(\d+)\s+Label: (\d+)\s+Prediction: (\d+)", out)]
        json.dump({"source": "reference plaintext flavour (make ptxt), nets/cifar/%s, NUM_SAMPLES=1" % net,
                   "progress_correct_image_label_prediction": progress},
                  open(os.path.join(HERE, "cifar_%s.json" % net), "w"), separators=(",", ":"))
        shutil.copyfile(os.path.join(REF, "nets/cifar", net, "var_prep.dat"), os.path.join(HERE, "cifar_%s_var_prep.dat" % net))
        print("cifar", net, progress)


if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    cifar()
    main()
    relu()
