#!/bin/bash
# Race / stress evidence for the hand-synchronised kernels (hand-counted s_waitcnt vmcnt, inline-asm LDS-DMA rings, the sibling-pacing spin
# on global progress words): the conv kernel tests (every gather / halo / stem / wgrad instantiation, fwd + dgrad + wgrad, fp32 + bf16) are
# run ROUNDS times under each of
#   - the shipped library,                                   pacing forced on for every sibling count (DIN_WGRAD_PACE=2)
#   - a build with the XCD tile remap off (-DDIN_XCD_REMAP=0: other workgroup -> tile placement, other L2 sharing), pacing forced on
#   - the shipped library with pacing off (DIN_WGRAD_PACE=0)
# while a second process keeps the device busy with an unrelated bandwidth-bound loop (uneven load: the case that exposes missing waits).
# The no-remap library is built on the build host:  tools/stress_gpu_suite.sh build   (into build_noremap/, which travels with gpurun)
#   usage on the GPU box: tools/stress_gpu_suite.sh run [ROUNDS] > profiles/rNN_stress_loop.log
set -u
cd "$(dirname "$0")/.."
CS=din-group-activity-recognition-benchmark_amd/csrc
if [ "${1:-run}" = build ]; then
  mkdir -p build_noremap
  FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DDIN_XCD_REMAP=0"
  for f in $CS/*.hip; do /opt/rocm/bin/hipcc $FL -c $f -o build_noremap/$(basename $f .hip).o 2>/dev/null & done
  /opt/rocm/bin/hipcc $FL -x hip -c $CS/din_error.cpp -o build_noremap/din_error.o
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_noremap/libdin_hip.so build_noremap/*.o
  ls -la build_noremap/libdin_hip.so
  exit 0
fi
if [ "${1:-run}" = build-experiments ]; then
  # the library WITH the timing knock-outs / experiment switches compiled in (DIN_GATHER_KNOCK, DIN_CONV_W16, DIN_CONV_RING): never shipped
  mkdir -p knock_build/experiments
  FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DDIN_EXPERIMENTS"
  for f in $CS/*.hip; do /opt/rocm/bin/hipcc $FL -c $f -o knock_build/experiments/$(basename $f .hip).o 2>/dev/null & done
  /opt/rocm/bin/hipcc $FL -x hip -c $CS/din_error.cpp -o knock_build/experiments/din_error.o
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o knock_build/experiments/libdin_hip.so knock_build/experiments/*.o
  ls -la knock_build/experiments/libdin_hip.so
  exit 0
fi
export DIN_OPTIONS_FROM_ENV=1      # the DIN_* variables below reach the library as options (ABI 8)
ROUNDS=${2:-3}
SEL="conv or wgrad or halo or gather or stem or dgrad or image"
python - <<'PY' &
import torch, time
x = torch.empty(1 << 28, device="cuda"); y = torch.empty_like(x)
t0 = time.time()
while time.time() - t0 < float(__import__("os").environ.get("STRESS_SECONDS", "900")):
    for _ in range(50):
        y.copy_(x); x.mul_(1.0001)
    torch.cuda.synchronize()
PY
NOISE=$!
fail=0
for r in $(seq 1 $ROUNDS); do
  for cfg in "shipped pace=2|DIN_WGRAD_PACE=2" "no-xcd-remap pace=2|DIN_WGRAD_PACE=2 DIN_LIB_PATH=$PWD/build_noremap/libdin_hip.so" "shipped pace=0|DIN_WGRAD_PACE=0"; do
    name=${cfg%%|*}; envs=${cfg##*|}
    out=$(env $envs python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "$SEL" -p no:cacheprovider 2>&1 | tail -1)
    echo "round $r  [$name]  $out"
    case "$out" in *failed*|*error*) fail=1;; esac
  done
done
kill $NOISE 2>/dev/null
echo "stress loop done: fail=$fail"
exit $fail
