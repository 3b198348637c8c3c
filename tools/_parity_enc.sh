python -m pytest tests/test_gpu_stages.py tests/test_gpu_codec.py -x -q -m gpu -k "encode or matches or ka" 2>&1 | tail -3
python tools/enc_only.py orig o1 occ3 base 2>&1 | tail -4
