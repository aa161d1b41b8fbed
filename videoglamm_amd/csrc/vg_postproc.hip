// Mask post-processing and evaluation counts on the device: integer / byte work, HBM-bound.
//
//   connected components : union-find over the pixel grid, all N images of a batch in ONE launch per pass (the
//                          reference's sam2/csrc/connected_components.cu:240-282 loops 6 launches per image).
//                          Passes: 64x64 tiles merged entirely in LDS (row runs from a ballot, one LDS atomicMin union
//                          per run contact, tile-local counts) -> unions across tile borders in global memory
//                          (3 % of the pixels) -> flatten + tile counts handed to the global roots -> finalize.
//                          The root of a component is its smallest pixel index, so labels are canonical:
//                          1 + min linear index.
//   remove_small_blobs   : 4-connectivity components smaller than min_size are cleared (R/eval_gcg_infer.py:20-29;
//                          skimage.morphology.remove_small_objects on a bool image).
//   fill_holes           : 8-connectivity background (score <= 0) components of area <= max_area get score 0.1
//                          (R/model/segment_anything_2/sam2/utils/misc.py:216-227).
//   mask_pair_counts     : |a & b|, |a | b| for every (prediction, ground truth) pair (R/eval_gcg_metrics.py:26-37,
//                          R/eval_referdavis_metrics.py:147-176).
//   boundary_counts      : the four integer counts of the boundary F-measure: seg2bmap of both masks, dilation by a
//                          disk, matches (R/eval_referdavis_metrics.py:194-305).
#include "vg_common.h"

namespace {

struct CCArgs {
  const void* in;     // uint8 mask (foreground = nonzero) or fp32 scores (foreground = score <= 0)
  int32_t* L;         // parent links / labels          [N, H, W]
  int32_t* cnt;       // per-root pixel counts / areas  [N, H, W]
  void* out;          // finalize modes 1, 2
  int N, H, W, conn;
  int mode;           // finalize: 0 labels+areas, 1 remove small (uint8 out), 2 fill holes (fp32 out)
  int limit;          // min_size (mode 1) / max_area (mode 2)
};

template <int IN>
__device__ __forceinline__ bool cc_fg(const void* in, int64_t i) {
  if (IN == 0) return ((const uint8_t*)in)[i] != 0;
  return ((const float*)in)[i] <= 0.f;
}

// ---- pass 1: a 64x64 tile per workgroup, merged completely in LDS.  A wave owns 16 rows (lane = column), so the runs
// of a row come from one ballot and there are no seams inside a tile.  Out: L[p] = global index of the pixel's TILE root
// (trees of depth 1), cnt[tile root] = size of the tile component, 0 elsewhere.
__device__ __forceinline__ int lds_find(volatile int* l, int a) {
  int q;
  while ((q = l[a]) != a) a = q;
  return a;
}
__device__ __forceinline__ void lds_unite(int* l, int a, int b) {
  while (true) {
    a = lds_find(l, a);
    b = lds_find(l, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&l[a], b);
    if (old == a) return;
    a = old;
  }
}
__device__ __forceinline__ bool bit64(uint64_t m, int i) { return (m >> i) & 1; }

template <int IN>
__global__ __launch_bounds__(256) void cc_tile_kernel(CCArgs p) {
  __shared__ int lbl[64 * 64];
  __shared__ int cntl[64 * 64];
  __shared__ uint64_t rowmask[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + lane, y0 = blockIdx.y * 64;
  const int64_t base = (int64_t)blockIdx.z * p.H * p.W;
  uint64_t bits[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave + 4 * i, y = y0 + r;
    const bool fg = x < p.W && y < p.H && cc_fg<IN>(p.in, base + (int64_t)y * p.W + x);
    const uint64_t b = __ballot(fg);
    bits[i] = b;
    if (lane == 0) rowmask[r] = b;
    int v = -1;
    if (fg) {
      const uint64_t below = lane ? (~b & ((1ull << lane) - 1)) : 0ull;   // background lanes left of this one
      v = r * 64 + (below ? 64 - __clzll((long long)below) : 0);          // first pixel of this pixel's run
    }
    lbl[r * 64 + lane] = v;
    cntl[r * 64 + lane] = 0;
  }
  __syncthreads();
  // a vertical contact is united once per pair of touching runs: when the left neighbour touches the same upper run
  // (left && ul) it has made (or delegated further left) the union
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave + 4 * i, idx = r * 64 + lane;
    if (r == 0 || !bit64(bits[i], lane)) continue;
    const uint64_t um = rowmask[r - 1];
    const bool left = lane > 0 && bit64(bits[i], lane - 1);
    const bool ul = lane > 0 && bit64(um, lane - 1);
    if (bit64(um, lane)) {
      if (!(left && ul)) lds_unite(lbl, idx, idx - 64);
    } else if (p.conn == 8) {
      if (ul && !left) lds_unite(lbl, idx, idx - 65);
      if (lane < 63 && bit64(um, lane + 1)) lds_unite(lbl, idx, idx - 63);
    }
  }
  __syncthreads();
  int roots[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave + 4 * i;
    bool active = bit64(bits[i], lane);
    const int root = active ? lds_find(lbl, r * 64 + lane) : -1;
    roots[i] = root;
    uint64_t m;
    while ((m = __ballot(active)) != 0) {                  // one LDS atomic per (row, component)
      const int leader = __ffsll((long long)m) - 1;
      const int rr = __shfl(root, leader, 64);
      const bool same = active && root == rr;
      const uint64_t sm = __ballot(same);
      if (lane == leader) atomicAdd(&cntl[rr], (int)__popcll(sm));
      active = active && !same;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave + 4 * i, y = y0 + r;
    if (x >= p.W || y >= p.H) continue;
    const int64_t g = base + (int64_t)y * p.W + x;
    const int root = roots[i];
    p.L[g] = root < 0 ? -1 : (y0 + (root >> 6)) * p.W + blockIdx.x * 64 + (root & 63);
    p.cnt[g] = root == r * 64 + lane ? cntl[root] : 0;
  }
}

// ---- pass 2: unions across tile borders (global memory).  Links are only ever lowered (atomicMin), the atomic returns the
// true previous value and a failed attempt continues with it: a stale plain load (the XCD L2s are not coherent with
// each other inside a kernel) costs a retry, never a wrong union.  Path halving keeps the walks short.
__device__ __forceinline__ int cc_find(int32_t* L, int a) {
  int q;
  while ((q = L[a]) != a) {
    const int g = L[q];
    if (g != q) L[a] = g;       // benign race: any ancestor is a valid link
    a = q;
  }
  return a;
}
__device__ __forceinline__ int cc_find_ro(const int32_t* L, int a) {   // no halving: pass 3 stores final roots concurrently
  int q;
  while ((q = L[a]) != a) a = q;
  return a;
}
__device__ __forceinline__ void cc_unite(int32_t* L, int a, int b) {
  while (true) {
    a = cc_find(L, a);
    b = cc_find(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;
  }
}

__global__ __launch_bounds__(192) void cc_border_kernel(CCArgs p) {
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;     // 0: top row, 1: left column, 2: right column
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64, W = p.W, H = p.H;
  int32_t* L = p.L + (int64_t)blockIdx.z * H * W;
  const int x = role == 0 ? x0 + lane : role == 1 ? x0 : x0 + 63;
  const int y = role == 0 ? y0 : y0 + lane;
  if (x >= W || y >= H) return;
  const int pix = y * W + x;
  if (L[pix] < 0) return;
  if (role == 0) {
    if (y == 0) return;
    const bool left = x > 0 && L[pix - 1] >= 0;
    const bool ul = x > 0 && L[pix - W - 1] >= 0;
    if (L[pix - W] >= 0) {
      if (!(lane > 0 && left && ul)) cc_unite(L, pix, pix - W);
    } else if (p.conn == 8) {
      if (ul && !(lane > 0 && left)) cc_unite(L, pix, pix - W - 1);
      if (x < W - 1 && L[pix - W + 1] >= 0) cc_unite(L, pix, pix - W + 1);
    }
  } else if (role == 1) {
    if (x == 0) return;
    if (L[pix - 1] >= 0) cc_unite(L, pix, pix - 1);
    // the diagonal into the left tile; row 0 of the tile belongs to the top-row threads.  With the upper pixel set the
    // contact is implied (upper ~ its own left neighbour by this same rule or by the run)
    else if (p.conn == 8 && lane > 0 && L[pix - W] < 0 && L[pix - W - 1] >= 0) cc_unite(L, pix, pix - W - 1);
  } else {
    if (p.conn != 8 || lane == 0 || x >= W - 1) return;
    if (L[pix - W] < 0 && L[pix - W + 1] >= 0) cc_unite(L, pix, pix - W + 1);
  }
}

// ---- pass 3: every pixel takes the global root; every tile root hands its tile count to the global root, tile roots of
// one tile that share a global root combined in an LDS table first (same-address device atomics cost ~90 ns each).
__global__ __launch_bounds__(256) void cc_flatten_count_kernel(CCArgs p) {
  constexpr int SLOTS = 64;
  __shared__ int hkey[SLOTS], hval[SLOTS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + lane, y0 = blockIdx.y * 64;
  const int64_t base = (int64_t)blockIdx.z * p.H * p.W;
  int32_t* L = p.L + base;
  int32_t* cnt = p.cnt + base;
  if (threadIdx.x < SLOTS) { hkey[threadIdx.x] = -1; hval[threadIdx.x] = 0; }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int y = y0 + wave + 4 * i;
    if (x >= p.W || y >= p.H) continue;
    const int pix = y * p.W + x;
    const int t = L[pix];
    if (t < 0) continue;
    const int root = cc_find_ro(L, t);
    L[pix] = root;              // concurrent walkers through this pixel see its old parent or the root: both ancestors
    const int c = cnt[pix];
    if (c > 0 && root != pix) {           // a tile root that is not the global root
      int s = (root * 0x9E3779B1u) >> 26;   // 6 bits
      bool done = false;
      for (int probe = 0; probe < SLOTS && !done; ++probe, s = (s + 1) & (SLOTS - 1)) {
        const int k = atomicCAS(&hkey[s], -1, root);
        if (k == -1 || k == root) { atomicAdd(&hval[s], c); done = true; }
      }
      if (!done) atomicAdd(&cnt[root], c);
    }
  }
  __syncthreads();
  if (threadIdx.x < SLOTS && hkey[threadIdx.x] >= 0) atomicAdd(&cnt[hkey[threadIdx.x]], hval[threadIdx.x]);
}

// one wave = 64 consecutive pixels of one row; a workgroup = 4 rows
template <int IN>
__global__ __launch_bounds__(256) void cc_finalize_kernel(CCArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + lane, y = blockIdx.y * 4 + wave;
  if (x >= p.W || y >= p.H) return;
  const int64_t base = (int64_t)blockIdx.z * p.H * p.W;
  const int64_t i = base + y * p.W + x;
  const int root = p.L[i];
  const bool fg = root >= 0;
  const int area = fg ? p.cnt[base + root] : 0;
  if (p.mode == 0) {
    p.L[i] = fg ? root + 1 : 0;
    p.cnt[i] = area;        // a global root rewrites its own value; everyone else only reads global roots
  } else if (p.mode == 1) {
    ((uint8_t*)p.out)[i] = (fg && area >= p.limit) ? 1 : 0;
  } else {
    ((float*)p.out)[i] = (fg && area <= p.limit) ? 0.1f : ((const float*)p.in)[i];
  }
}

template <int IN>
int cc_run(const CCArgs& p, hipStream_t st) {
  const dim3 tiles((p.W + 63) / 64, (p.H + 63) / 64, p.N);
  cc_tile_kernel<IN><<<tiles, 256, 0, st>>>(p);
  if (tiles.x > 1 || tiles.y > 1) cc_border_kernel<<<tiles, 192, 0, st>>>(p);
  cc_flatten_count_kernel<<<tiles, 256, 0, st>>>(p);
  cc_finalize_kernel<IN><<<dim3((p.W + 63) / 64, (p.H + 3) / 4, p.N), 256, 0, st>>>(p);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

int cc_check(const void* in, const void* a, const void* b, int N, int H, int W, const char* who) {
  VG_CHECK(in && a && b, VG_ERR_ARG, "%s: null pointer", who);
  VG_CHECK(N > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && N <= 65535, VG_ERR_ARG, "%s: bad shape N=%d H=%d W=%d", who, N, H, W);
  return VG_OK;
}

// ---------------------------------------------------------------------------------------------- pair counts
__device__ __forceinline__ uint32_t nz_bytes(uint32_t v) {   // 0x80 in every byte of v that is nonzero
  return (v | ((v & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
}

__global__ __launch_bounds__(256) void pair_counts_kernel(const uint8_t* a, const uint8_t* b, unsigned long long* inter,
                                                          unsigned long long* uni, int G, int64_t L, int vec, int diag) {
  const int pair = blockIdx.y, pi = diag ? pair : pair / G, gi = diag ? pair : pair % G;
  const uint8_t* pa = a + (int64_t)pi * L;
  const uint8_t* pb = b + (int64_t)gi * L;
  unsigned ci = 0, cu = 0;       // a thread sees at most L / (gridDim.x * 256) * 16 < 2^32 pixels
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  if (vec) {
    const int64_t nw = L / 16;
    for (int64_t w = tid; w < nw; w += nth) {
      const u32x4_t va = __builtin_nontemporal_load((const u32x4_t*)pa + w);
      const u32x4_t vb = __builtin_nontemporal_load((const u32x4_t*)pb + w);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t na = nz_bytes(va[k]), nb = nz_bytes(vb[k]);
        ci += __popc(na & nb);
        cu += __popc(na | nb);
      }
    }
    for (int64_t i = nw * 16 + tid; i < L; i += nth) {
      const bool x = pa[i] != 0, y = pb[i] != 0;
      ci += x && y;
      cu += x || y;
    }
  } else {
    for (int64_t i = tid; i < L; i += nth) {
      const bool x = pa[i] != 0, y = pb[i] != 0;
      ci += x && y;
      cu += x || y;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ci += __shfl_xor(ci, o, 64);
    cu += __shfl_xor(cu, o, 64);
  }
  // one atomic pair per workgroup: same-address device atomics serialise at ~90 ns each on this part
  __shared__ unsigned red[2][4];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ci; red[1][threadIdx.x >> 6] = cu; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long ti = (unsigned long long)red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const unsigned long long tu = (unsigned long long)red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (ti) atomicAdd(&inter[pair], ti);
    if (tu) atomicAdd(&uni[pair], tu);
  }
}

// ---------------------------------------------------------------------------------------------- boundary F counts
// 64x64 output pixels per workgroup, everything as ROW BITMASKS in LDS (a row of the 64 + 2r (+1) wide halo tile is
// 128 bits): the masks are staged with one ballot per 64 columns, the boundary stencil of _seg2bmap is three shifts and
// xors per row, and the disk test of a pixel is one shifted window compare per disk row (2r + 1 of them instead of
// ~3r^2 byte reads).  Only pixels that lie on one of the two boundaries search at all.
struct Bits128 {
  uint64_t lo, hi;
};
__device__ __forceinline__ Bits128 shr1(Bits128 v) { return {(v.lo >> 1) | (v.hi << 63), v.hi >> 1}; }
__device__ __forceinline__ uint64_t window64(Bits128 v, int sh) {   // bits [sh, sh + 63] of v, 0 <= sh < 128
  if (sh >= 64) return v.hi >> (sh - 64);
  return sh ? (v.lo >> sh) | (v.hi << (64 - sh)) : v.lo;
}

__global__ __launch_bounds__(256) void boundary_counts_kernel(const uint8_t* fg, const uint8_t* gt, unsigned long long* out,
                                                              int H, int W, int r) {
  __shared__ Bits128 seg[2][128];     // [mask][row]: rows y0 .. y0 + BW (one extra row for the stencil)
  __shared__ Bits128 bnd[2][128];     // boundary maps, rows y0 .. y0 + BW - 1
  __shared__ int hw[64];              // half-width of the disk row dy = i - r
  __shared__ unsigned red[4][4];
  __shared__ uint16_t list[64 * 64];  // compacted boundary pixels: lx | ly << 7 | on-fg << 14 | on-gt << 15
  __shared__ int nlist;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) nlist = 0;
  const int BW = 64 + 2 * r, SW = BW + 1;
  const int x0 = blockIdx.x * 64 - r, y0 = blockIdx.y * 64 - r;
  const int64_t base = (int64_t)blockIdx.z * H * W;
  for (int j0 = wave * 8; j0 < 4 * SW; j0 += 32) {           // jobs = (mask, row, 64-column half); 8 loads in flight per lane
    bool v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int job = j0 + k, m = job & 1, half = (job >> 1) & 1, row = job >> 2;
      const int yy = y0 + row, xx = x0 + half * 64 + lane;
      const uint8_t* src = m ? gt : fg;
      v[k] = job < 4 * SW && yy >= 0 && yy < H && xx >= 0 && xx < W && (half * 64 + lane) < SW && src[base + (int64_t)yy * W + xx] != 0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int job = j0 + k, m = job & 1, half = (job >> 1) & 1, row = job >> 2;
      const uint64_t b = __ballot(v[k]);
      if (lane == 0 && job < 4 * SW) { if (half) seg[m][row].hi = b; else seg[m][row].lo = b; }
    }
  }
  if (tid <= 2 * r) {
    const int dy = tid - r;
    int w = 0;
    while ((w + 1) * (w + 1) + dy * dy <= r * r) ++w;
    hw[tid] = w;
  }
  __syncthreads();
  if (tid < 2 * BW) {                                          // one thread per (mask, row) of the boundary maps
    const int m = tid & 1, row = tid >> 1, yy = y0 + row;
    Bits128 b = {0, 0};
    if (yy >= 0 && yy < H) {
      const Bits128 S = seg[m][row], N = seg[m][row + 1];      // N is all zero below the image
      const Bits128 e = shr1(S), se = shr1(N);
      if (yy == H - 1) b = {S.lo ^ e.lo, S.hi ^ e.hi};
      else b = {(S.lo ^ e.lo) | (S.lo ^ N.lo) | (S.lo ^ se.lo), (S.hi ^ e.hi) | (S.hi ^ N.hi) | (S.hi ^ se.hi)};
      const int lc = W - 1 - x0;                               // the image's last column: seg ^ south, corner 0
      if (lc >= 0 && lc < 128) {
        const uint64_t bit = 1ull << (lc & 63);
        const uint64_t v = yy == H - 1 ? 0ull : ((lc < 64 ? S.lo ^ N.lo : S.hi ^ N.hi) & bit);
        if (lc < 64) b.lo = (b.lo & ~bit) | v; else b.hi = (b.hi & ~bit) | v;
      }
      // keep only columns inside the image: bits [max(0, -x0), min(127, W - 1 - x0)]
      const int c0 = x0 < 0 ? -x0 : 0, c1 = lc < 127 ? lc : 127;
      Bits128 keep = {0, 0};
      if (c1 >= c0) {
        const uint64_t lo_m = c0 < 64 ? (~0ull << c0) : 0ull, hi_m = c0 < 64 ? ~0ull : (~0ull << (c0 - 64));
        const uint64_t lo_M = c1 < 64 ? (~0ull >> (63 - c1)) : ~0ull, hi_M = c1 < 64 ? 0ull : (~0ull >> (127 - c1));
        keep = {lo_m & lo_M, hi_m & hi_M};
      }
      b = {b.lo & keep.lo, b.hi & keep.hi};
    }
    bnd[m][row] = b;
  }
  __syncthreads();
  // boundary pixels are a few per cent of the tile: compact them into a list first so that the disk search runs with
  // full waves (a wave that holds a single boundary pixel would otherwise walk all 2r + 1 disk rows for it)
  unsigned nf = 0, ng = 0, mf = 0, mg = 0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int ly = r + wave + 4 * i, lx = r + lane;
    const bool f = (window64(bnd[0][ly], lx) & 1) != 0, g = (window64(bnd[1][ly], lx) & 1) != 0;
    const uint64_t m = __ballot(f || g);
    if (m == 0) continue;                                   // wave-uniform
    int slot = 0;
    if (lane == 0) slot = atomicAdd(&nlist, (int)__popcll(m));
    slot = __shfl(slot, 0, 64) + (int)__popcll(m & ((1ull << lane) - 1));
    if (f || g) list[slot] = (uint16_t)(lx | (ly << 7) | (f ? 0x4000 : 0) | (g ? 0x8000 : 0));
    nf += f;
    ng += g;
  }
  __syncthreads();
  const int n = nlist;
  for (int e = tid; e < n; e += 256) {
    const int v = list[e], lx = v & 127, ly = (v >> 7) & 127;
    const bool f = v & 0x4000, g = v & 0x8000;
    bool df = false, dg = false;          // the OTHER map's boundary within the disk of this pixel
    for (int dy = -r; dy <= r; ++dy) {
      const int w = hw[dy + r];
      const uint64_t win = w >= 31 ? ~0ull >> 1 : (1ull << (2 * w + 1)) - 1;    // 2w + 1 <= 63 bits
      if (g) df |= (window64(bnd[0][ly + dy], lx - w) & win) != 0;
      if (f) dg |= (window64(bnd[1][ly + dy], lx - w) & win) != 0;
    }
    mf += f && dg;      // fg boundary pixels matched by the dilated gt boundary
    mg += g && df;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    nf += __shfl_xor(nf, o, 64);
    ng += __shfl_xor(ng, o, 64);
    mf += __shfl_xor(mf, o, 64);
    mg += __shfl_xor(mg, o, 64);
  }
  if (lane == 0) { red[0][wave] = nf; red[1][wave] = ng; red[2][wave] = mf; red[3][wave] = mg; }
  __syncthreads();
  if (tid < 4) {                          // one atomic per counter per workgroup
    const unsigned t = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    if (t) atomicAdd(out + (int64_t)blockIdx.z * 4 + tid, (unsigned long long)t);
  }
}

}  // namespace

extern "C" int vg_connected_components(const uint8_t* mask, int32_t* labels, int32_t* counts, int N, int H, int W,
                                       int connectivity, vg_stream_t stream) {
  if (int e = cc_check(mask, labels, counts, N, H, W, "vg_connected_components")) return e;
  VG_CHECK(connectivity == 4 || connectivity == 8, VG_ERR_ARG, "vg_connected_components: connectivity %d not in {4, 8}", connectivity);
  CCArgs p{mask, labels, counts, nullptr, N, H, W, connectivity, 0, 0};
  return cc_run<0>(p, (hipStream_t)stream);
}

extern "C" int vg_remove_small_blobs(const uint8_t* mask, uint8_t* out, int32_t* ws_labels, int32_t* ws_counts, int N,
                                     int H, int W, int min_size, vg_stream_t stream) {
  if (int e = cc_check(mask, ws_labels, ws_counts, N, H, W, "vg_remove_small_blobs")) return e;
  VG_CHECK(out, VG_ERR_ARG, "vg_remove_small_blobs: null output");
  CCArgs p{mask, ws_labels, ws_counts, out, N, H, W, 4, 1, min_size};
  return cc_run<0>(p, (hipStream_t)stream);
}

extern "C" int vg_fill_holes(const float* scores, float* out, int32_t* ws_labels, int32_t* ws_counts, int N, int H,
                             int W, int max_area, vg_stream_t stream) {
  if (int e = cc_check(scores, ws_labels, ws_counts, N, H, W, "vg_fill_holes")) return e;
  VG_CHECK(out, VG_ERR_ARG, "vg_fill_holes: null output");
  VG_CHECK(max_area > 0, VG_ERR_ARG, "vg_fill_holes: max_area must be positive");   // misc.py:222
  CCArgs p{scores, ws_labels, ws_counts, out, N, H, W, 8, 2, max_area};
  return cc_run<1>(p, (hipStream_t)stream);
}

extern "C" int vg_mask_pair_counts(const uint8_t* a, const uint8_t* b, int64_t* inter, int64_t* uni, int P, int G,
                                   int64_t L, int diagonal, vg_stream_t stream) {
  VG_CHECK(a && b && inter && uni, VG_ERR_ARG, "vg_mask_pair_counts: null pointer");
  VG_CHECK(P > 0 && G > 0 && L > 0, VG_ERR_ARG, "vg_mask_pair_counts: bad shape P=%d G=%d L=%lld", P, G, (long long)L);
  VG_CHECK(!diagonal || P == G, VG_ERR_ARG, "vg_mask_pair_counts: diagonal needs P == G (got %d, %d)", P, G);
  const int64_t pairs = diagonal ? P : (int64_t)P * G;
  VG_CHECK(pairs <= 65535, VG_ERR_UNSUPPORTED, "vg_mask_pair_counts: %lld pairs > 65535", (long long)pairs);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(inter, 0, sizeof(int64_t) * pairs, st) != hipSuccess || hipMemsetAsync(uni, 0, sizeof(int64_t) * pairs, st) != hipSuccess) {
    vg_set_error("vg_mask_pair_counts: memset failed");
    return VG_ERR_LAUNCH;
  }
  const int vec = (L % 16 == 0) && ((((uintptr_t)a | (uintptr_t)b) & 15) == 0);
  int chunks = (int)((L / 16 + 255) / 256);      // at most 64 workgroups (= atomics per counter) per pair, >= 4 KB each
  chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
  pair_counts_kernel<<<dim3(chunks, (unsigned)pairs), 256, 0, st>>>(a, b, (unsigned long long*)inter, (unsigned long long*)uni, G, L, vec, diagonal);
  VG_LAUNCH_CHECK();
  return VG_OK;
}

extern "C" int vg_boundary_counts(const uint8_t* fg, const uint8_t* gt, int64_t* out, int N, int H, int W, int radius,
                                  vg_stream_t stream) {
  VG_CHECK(fg && gt && out, VG_ERR_ARG, "vg_boundary_counts: null pointer");
  VG_CHECK(N > 0 && N <= 65535 && H > 0 && W > 0, VG_ERR_ARG, "vg_boundary_counts: bad shape N=%d H=%d W=%d", N, H, W);
  VG_CHECK(radius >= 0 && radius <= 31, VG_ERR_UNSUPPORTED, "vg_boundary_counts: radius %d not in [0, 31]", radius);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(int64_t) * 4 * N, st) != hipSuccess) {
    vg_set_error("vg_boundary_counts: memset failed");
    return VG_ERR_LAUNCH;
  }
  boundary_counts_kernel<<<dim3((W + 63) / 64, (H + 63) / 64, N), 256, 0, st>>>(fg, gt, (unsigned long long*)out, H, W, radius);
  VG_LAUNCH_CHECK();
  return VG_OK;
}
