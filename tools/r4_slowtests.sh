T="tests/test_gpu_codec.py::test_truncated_codestream_decodes_like_oracle tests/test_gpu_codec.py::test_corrupted_block_bytes_decode_like_oracle tests/test_gpu_codec.py::test_sample_containers_8_16_32_agree"
echo NEW; timeout 400 python -m pytest $T -m gpu -q --durations=3 2>&1 | tail -8
cp openjph_amd/libojphgpu.so /tmp/keep.so; cp openjph_amd/variants/lib_old.so openjph_amd/libojphgpu.so
echo OLD; timeout 400 python -m pytest $T -m gpu -q --durations=3 2>&1 | tail -6
cp /tmp/keep.so openjph_amd/libojphgpu.so
