#!/bin/bash
# the whole -m gpu suite -> gpurun_out/$OUT/tests.txt (default r04_tests); with "nocache" as first argument also the kernel / module / golden /
# full-size files once more with PYTORCH_NO_CUDA_MEMORY_CACHING=1 (every tensor its own hipMalloc: an access past an
# operand is then far more likely to leave the mapping -- the closest thing to a sanitiser on this pool; graph tests need
# the caching allocator and are left out)
set -u
O=gpurun_out/${OUT:-r04_tests}; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -15 $O/tests.txt
if [ "${1:-}" = "nocache" ]; then
  PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py tests/test_prod_gpu.py \
    tests/test_fullsize_gpu.py -m gpu -q -x -k 'not graph' > $O/tests_nocache.txt 2>&1; echo "nocache rc=$?"; tail -3 $O/tests_nocache.txt
fi
