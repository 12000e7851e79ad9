#!/bin/bash
# builds tools/probes/line_probe (standalone harness of csrc/conv_line.hip) for gfx950; prints the kernels' register use
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value \
  -Rpass-analysis=kernel-resource-usage -o line_probe line_probe.hip 2>&1 | grep -E "error|conv_line.*(Name|VGPRs:|Scratch|SGPRs Spill)" | sed -e 's/.*remark: //' | head -40
