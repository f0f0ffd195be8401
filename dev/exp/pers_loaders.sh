# loader wavefronts per workgroup (2 / 4 / 8) x ablation bits on the persistent GEMM
for nl in 2 4 8; do for d in 0 8 10; do
  echo "== loaders $nl BOFI_GEMM_DBG=$d"
  BOFI_GEMM_PERS_LOADERS=$nl BOFI_GEMM_DBG=$d python dev/exp/mb_pers.py 11520x2048x512xln 11520x6144x512xln 2>&1 | grep -v amdgpu.ids
done; done
