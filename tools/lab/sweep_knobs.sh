run() { echo "== $*"; env "$@" python tools/bench_parts.py 3 hiera 2>&1 | tail -1; env "$@" python tools/bench_parts.py 3 prefill 2>&1 | tail -1; }
run VG_X=0
run VG_GEMM_NT_MB=64
run VG_GEMM_NT_MB=256
run VG_GEMM_GN=2
run VG_GEMM_GN=8
run VG_W128_MINKB=4608
run VG_ATTN_NW8=0
run VG_X=0
