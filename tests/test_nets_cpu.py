"""Host logic around the bootstraps, pinned against the reference's own plaintext build:
weight-file parsing (redsec_amd/nets.py) + layer index math reproduce the logits that
`make ptxt` prints for nets/mnist/sign1024x1 on all 100 bundled images, and the accuracies /
every-10th predictions of sign1024x2 and sign1024x3."""
import json
import os

import numpy as np
import pytest

import plain_model as pm


def test_sign1024x1_logits_equal_reference_plaintext():
    gold = json.load(open(os.path.join(pm.GOLD, "mnist_sign1024x1.json")))
    labels, pixels = pm.load_images()
    net = pm.load_net("sign1024x1")
    correct = 0
    for i in range(100):
        logits = pm.forward(net, pixels[i])
        assert logits.tolist() == gold["logits"][i], i
        correct += int(np.argmax(logits) == labels[i])
    assert correct == round(gold["accuracy_percent"])          # 96/100
    assert gold["logits"][0] == [-109, -151, -32, 70, -258, 86, -127, -114, -30, -81]  # BASELINE.md known answer


@pytest.mark.parametrize("name", ["sign1024x2", "sign1024x3"])
def test_deeper_nets_match_reference_predictions(name):
    gold = json.load(open(os.path.join(pm.GOLD, "mnist_%s.json" % name)))
    labels, pixels = pm.load_images()
    net = pm.load_net(name)
    preds = np.array([int(np.argmax(pm.forward(net, pixels[i]))) for i in range(100)])
    assert int((preds == labels).sum()) == round(gold["accuracy_percent"])
    for correct, image_i, label, pred in gold["progress_correct_image_label_prediction"]:
        assert labels[image_i] == label and preds[image_i] == pred
        assert int((preds[:image_i + 1] == labels[:image_i + 1]).sum()) == correct


def test_weight_file_layout_sign1024x1():
    # 56,881 B = 5 + 50,177 + 4,097 + 2,561 + 41
    blob = open(os.path.join(pm.GOLD, "mnist_sign1024x1_var_prep.dat"), "rb").read()
    assert len(blob) == 56881
    net = pm.load_net("sign1024x1")
    assert net.fc[0][0].shape == (196, 1024) and net.final[0].shape == (1024, 10)
    assert set(np.unique(net.fc[0][1])) <= {0, 1}


def test_cifar_checker_matches_reference_prediction():
    """The CIFAR plaintext checker (conv / max-pool / FC index math) reproduces the prediction the
    reference's own `make ptxt` build prints for its one sample (NUM_SAMPLES = 1, main.cpp:25)."""
    labels, pix = pm.load_cifar_images()
    gold = json.load(open(os.path.join(pm.GOLD, "cifar_binarynet_small.json")))
    (_, image_i, label, pred), = gold["progress_correct_image_label_prediction"]
    net = pm.CifarNet("binarynet_small")
    assert labels[image_i] == label
    assert int(np.argmax(pm.cifar_forward(net, pix[image_i]))) == pred
