# ablations of the LDS-DMA GEMM on large-M shapes: BOFI_GEMM_DBG bits 1 = no global->LDS loads, 2 = no LDS reads / MFMA, 8 = no epilogue
for sh in 9216 2048 512 -- 9216 512 2048 -- 9216 1536 512; do :; done
for args in "9216 2048 512" "9216 512 2048" "2304 2048 512"; do
  for d in 0 16 17 18 19 24 27; do BOFI_GEMM_DBG=$d python dev/mb_graph2.py $args 2>&1 | tail -1; done
done
