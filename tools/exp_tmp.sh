tools/gpu_pmc.sh r1b_fetch lstm0 "FETCH_SIZE" > gpurun_out/r1b_pmc_fetch.txt 2>&1
tools/gpu_pmc.sh r1b_tcc lstm0 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" > gpurun_out/r1b_pmc_tcc.txt 2>&1
grep -A3 lstm16 gpurun_out/r1b_pmc_fetch.txt; grep -A4 lstm16 gpurun_out/r1b_pmc_tcc.txt
