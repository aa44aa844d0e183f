for d in 0 1 2 3 4 5 7; do echo "== dbg=$d (1=no compute 2=no stage 4=no reduce)"; VMX_DBG=$d timeout -k 10 100 python -u tools/gemm_floor.py | tail -1; done
