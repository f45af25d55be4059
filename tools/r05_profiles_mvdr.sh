#!/bin/bash
# round 5: PMC traffic and SQ passes of the MVDR kernels (BASELINE configs[3]) -- VERDICT r4 #5
bash tools/pmc_traffic.sh adaptive gpurun_out/pmc_traffic_mvdr "--config mvdr" > gpurun_out/pmc_traffic_mvdr.log 2>&1
bash tools/pmc_sq.sh adaptive gpurun_out/pmc_sq_mvdr "--config mvdr" > gpurun_out/pmc_sq_mvdr.log 2>&1
tail -30 gpurun_out/pmc_traffic_mvdr.log
