#!/bin/bash
# in-situ passes over the secondary workloads with the current kernels: the upscale UNet (B = 4, 64x64) and the VAE decoder
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
INSITU_KIND=upscale INSITU_B=4 INSITU_TOPK=10 timeout 1500 python scripts/tune_insitu.py 64 64 gpurun_out/tuned_ups.json 2>&1 | grep -v "^\[" | tee gpurun_out/insitu_ups.log | tail -25
UPGPT_TUNE_FILE=$R/gpurun_out/tuned_ups.json INSITU_VAE=1 INSITU_TOPK=8 timeout 1500 python scripts/tune_insitu.py 32 32 gpurun_out/tuned_ups_vae.json 2>&1 | grep -v "^\[" | tee gpurun_out/insitu_vae.log | tail -25
