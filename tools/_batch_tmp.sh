python -m pytest tests/test_cm_gpu.py tests/test_replicas_gpu.py tests/test_bench_contract_gpu.py tests/test_flood_gpu.py -m gpu -q -s 2>&1 | tail -25
C=ecc_ldpc_amd/ecc-ldpc-hip
for hc in 1,0 8,0 8,8 64,0 64,64 256,64; do $C 3.2 ldpc/hip-minsum/jpl.4096.4.5/50/4/5 -m20000 -H$hc -ccodes 2>&1 | grep harness-visible; done
for hc in 1,0 64,64; do $C 2.0 ldpc/hip-minsum/jpl.4096.4.5/50/4/5 -m8000 -H$hc -ccodes 2>&1 | grep harness-visible; done
