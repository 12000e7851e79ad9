timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "amdgpu.ids\|Inference\|Deactivate"
DIN_CONV_KORDER=0 timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "amdgpu.ids\|Inference\|Deactivate"
