#!/bin/bash
OUT=gpurun_out/r03_dbg; mkdir -p $OUT
python - <<'PY' > $OUT/dbg.txt 2>&1
import os, sys, shutil, subprocess, tempfile
ROOT=os.getcwd(); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import plain_model as pm, refdrivers as rd
net_name="binarynet_small"
tmp=tempfile.mkdtemp()
client=os.path.join(tmp,"client"); netdir=os.path.join(tmp,"nets","cifar",net_name)
os.makedirs(client); os.makedirs(netdir)
shutil.copyfile(os.path.join(rd.GOLD,"cifar_%s_var_prep.dat"%net_name), os.path.join(netdir,"var_prep.dat"))
print(rd.run("client_gen_secure_keyset.out", client).returncode)
labels,pix=pm.load_cifar_images()
open(os.path.join(client,"img.csv"),"w").write(",".join(str(int(v)) for v in [labels[1],32,32,3]+list(pix[1]))+",\n")
print(rd.run("client_encrypt_image.out", client, "img.csv").returncode)
env=dict(os.environ); env["LD_LIBRARY_PATH"]=os.path.join(ROOT,"redsec_amd")+":"+env.get("LD_LIBRARY_PATH","")
exe=os.path.join(rd.REFNETS,"cifar_%s_enc.out"%net_name)
r=subprocess.run(["/opt/rocm/bin/rocgdb","-batch","-ex","run","-ex","bt","--args",exe],cwd=netdir,env=env,capture_output=True,text=True,timeout=400)
print(r.stdout[-5000:]); print(r.stderr[-2000:])
PY
tail -n 80 $OUT/dbg.txt
