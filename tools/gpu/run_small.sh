timeout 900 python -m pytest tests/test_gpu_din_model.py -x -q -k "sibling" 2>&1 | tail -8
