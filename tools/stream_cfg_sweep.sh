run() { env "$@" python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), 'single', round(d['single_stream_frames_per_s'],1))"; }
run A=1
run LWP_GEMM_C3=32,64,2
run LWP_GEMM_C3=64,64,1
run LWP_GEMM_C3=64,64,2
run LWP_GEMM_C3=32,32,4
run LWP_GEMM_PW=32,64,2
run LWP_GEMM_C3=32,64,2 LWP_GEMM_PW=32,64,2
run LWP_DWPW_NW=8
run LWP_DWPW_NW=4
run LWP_GEMM_C3=32,64,2 LWP_DWPW_NW=8
