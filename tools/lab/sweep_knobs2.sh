c1() { echo "== C1 $*"; env "$@" python bench.py --frames 8 --te 8 --src 512 --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | cut -c60-150; }
c2() { echo "== C2 $*"; env "$@" python bench.py --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | cut -c60-150; }
vid() { echo "== C2 video $*"; env "$@" python bench.py --branch video --no-cpu-baseline --no-quality --no-roofline 2>/dev/null | cut -c60-150; }
c1 VG_X=0; c1 VG_DEC_KPW_MIN=1024; c1 VG_X=0; c1 VG_DEC_KPW_MIN=1024
c2 VG_X=0; c2 VG_GEMM_SPLITK_TILES=128; c2 VG_GEMM_SPLITK_TILES=512; c2 VG_GEMM_TAILSPLIT=0; c2 VG_X=0
vid VG_X=0; vid VG_ATTN_SPLIT_WG_D256=128; vid VG_ATTN_SPLIT_WG_D256=512; vid VG_GEMM_SPLITK_TILES=512; vid VG_X=0
