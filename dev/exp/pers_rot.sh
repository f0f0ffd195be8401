for d in 0 32 8 40 10 42; do
  echo "== BOFI_GEMM_DBG=$d"
  BOFI_GEMM_DBG=$d python dev/exp/mb_pers.py 11520x2048x512xln 11520x6144x512xln 11520x512x2048xres 2>&1 | grep -v amdgpu.ids
done
