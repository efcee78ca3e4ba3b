#!/bin/bash
# Shader clock and package power (rocm-smi, read-only) while (a) the library's BN254 MSM 2^24 with tables and (b) the pure-arithmetic XYZZ
# micro-benchmark run in a loop: what "sustained clock" means for the issue-rate figures in DESIGN.md.  Run on the GPU box from the repo root.
sample() { for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/^[^:]*: *//' | paste -sd' ' ; sleep 1; done; }
python - <<'PY' &
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from panda_amd import gpu_ffi as ffi
lib = ffi.load(); dev = torch.device("cuda", 0)
log_n = 24; n = 1 << log_n
st = torch.cuda.Stream(device=dev); ps = ffi.PandaStream(st.cuda_stream)
b = torch.empty(n * 64, dtype=torch.uint8, device=dev); s = torch.empty(n * 32, dtype=torch.uint8, device=dev); r = torch.zeros(96, dtype=torch.uint8, device=dev)
ffi.check(lib.panda_gen_bases(0, 1, 0, n, b.data_ptr(), ps), "g"); ffi.check(lib.panda_gen_scalars(0, 2, 0, n, s.data_ptr(), ps), "g")
ffi.check(lib.panda_msm_precompute_bases(0, b.data_ptr(), log_n, 0, ps), "p")
cfg = ffi.MSMConfiguration(ffi.PandaMemPool(), ps, b.data_ptr(), s.data_ptr(), r.data_ptr(), log_n, 0)
t = time.time()
while time.time() - t < 12: ffi.check(lib.panda_msm_execute_bn254(cfg), "m")
PY
sleep 7
echo "--- during panda_msm_execute_bn254 (2^24, tables) in a loop:"
sample
wait
if [ -x gpurun_out/ubench_batchaffine ]; then
  ( end=$((SECONDS+10)); while [ $SECONDS -lt $end ]; do ./gpurun_out/ubench_batchaffine > /dev/null; done ) &
  sleep 4
  echo "--- during tools/ubench_batchaffine.hip (arithmetic only, no HBM traffic) in a loop:"
  sample
  wait
fi
echo "--- idle:"
sample
