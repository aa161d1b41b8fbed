run() { echo "== $*"; env "$@" python tools/bench_parts.py 3 towers 2>&1 | tail -1; }
run VG_X=0; run VG_GEMM_S128=0; run VG_GEMM_S128=2; run VG_GEMM_TAILSPLIT=0; run VG_GEMM_SPLITK=0; run VG_TOWERS_OVERLAP=0; run VG_GEMM_W128=0; run VG_X=0
