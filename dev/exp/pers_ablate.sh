# Ablation of the persistent GEMM (BOFI_GEMM_DBG bits: 1 no loads, 2 no LDS reads / MFMA, 8 no epilogue) on the large shapes
for d in 0 1 2 8 3 9 10 11; do
  echo "== BOFI_GEMM_DBG=$d"
  BOFI_GEMM_DBG=$d python dev/exp/mb_pers.py 11520x2048x512xln 11520x512x2048xres 5760x1536x512xln 2>&1 | grep -v amdgpu.ids
done
