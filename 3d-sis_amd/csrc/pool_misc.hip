// MaxPool3d(3,1,1) and layout helpers, channels-last, gfx950.
//
// Replaces nn.MaxPool3d(3,1,1) (lib/nets/backbones.py:206,210,220).  With channels-last
// activations a voxel is one contiguous C-float row, so each of the 27 taps is a fully
// coalesced 16 B/lane read served by L1/L2 (the maps are <= 3.5 MB); out-of-range taps are
// skipped, which is the -inf padding of the reference.
#include "common.h"
#include <float.h>
#include <stdlib.h>

namespace {

// thread = (x, y, z-segment of ZSEG voxels, 4 channels): the 3x3 (x,y) column maxima of ZSEG+2 consecutive z are
// computed once and reused by three outputs each -> 9*(ZSEG+2)/ZSEG loads per output instead of 27.  The maps of this
// network are small (6912 voxels x 16..32 float4 channels), so the launcher trades that reuse for parallelism
// (ZSEG = 8 on a 24x12x24x128 map left 3/4 of the CUs idle): ZSEG 2 on maps with >= 200k (voxel, float4) pairs, else 1.
template <int ZSEG>
__global__ __launch_bounds__(256) void maxpool3_kernel(const float4 *__restrict__ in, int X, int Y, int Z, int C4,
                                                       float4 *__restrict__ out, int O4)
{
    const int nseg = (Z + ZSEG - 1) / ZSEG;
    const int64_t total = (int64_t)X * Y * nseg * C4;
    const float ninf = -__builtin_huge_valf();
    // XCD-aware block order (block b runs on XCD b % 8, each with its own L2): every XCD takes one contiguous range of the
    // x-major work list, i.e. a slab of x-planes, so a voxel's 27 taps hit the L2 that already holds its neighbours
    // (PMC, round 2: 21-125 MB fetched for 3.5-14 MB maps when consecutive blocks alternate XCDs)
    int64_t vb;
    {
        const int nb = gridDim.x, xcd = blockIdx.x % 8, idx = blockIdx.x / 8, qd = nb / 8, rm = nb % 8;
        vb = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    for (int64_t t = vb * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C4);
        int64_t v = t / C4;
        const int sg = (int)(v % nseg);
        v /= nseg;
        const int y = (int)(v % Y), x = (int)(v / Y);
        const int z0 = sg * ZSEG;
        float4 col[ZSEG + 2];
#pragma unroll
        for (int k = 0; k < ZSEG + 2; ++k) col[k] = make_float4(ninf, ninf, ninf, ninf);
        // one x-slice of the window at a time: its 3 x (ZSEG+2) taps are requested together (clamped addresses, no
        // branches), then folded in with out-of-range taps replaced by -inf.  The branchy tap-by-tap form compiled to a
        // load -> s_waitcnt vmcnt(0) chain of 27..54 L1/L2 latencies per thread.
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            const bool vx = xx >= 0 && xx < X;
            const int xc = min(max(xx, 0), X - 1);
            float4 f[3][ZSEG + 2];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int yc = min(max(y + dy, 0), Y - 1);
                const float4 *row = in + (((int64_t)xc * Y + yc) * Z) * C4 + c;
#pragma unroll
                for (int k = 0; k < ZSEG + 2; ++k) {
                    const int zc = min(max(z0 - 1 + k, 0), Z - 1);
                    f[dy + 1][k] = row[(int64_t)zc * C4];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                const bool vxy = vx && yy >= 0 && yy < Y;
#pragma unroll
                for (int k = 0; k < ZSEG + 2; ++k) {
                    const int zz = z0 - 1 + k;
                    const bool ok = vxy && zz >= 0 && zz < Z;
                    const float4 t = f[dy + 1][k];
                    col[k].x = fmaxf(col[k].x, ok ? t.x : ninf); col[k].y = fmaxf(col[k].y, ok ? t.y : ninf);
                    col[k].z = fmaxf(col[k].z, ok ? t.z : ninf); col[k].w = fmaxf(col[k].w, ok ? t.w : ninf);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < ZSEG; ++j) {
            const int z = z0 + j;
            if (z >= Z) break;
            float4 m;
            m.x = fmaxf(fmaxf(col[j].x, col[j + 1].x), col[j + 2].x);
            m.y = fmaxf(fmaxf(col[j].y, col[j + 1].y), col[j + 2].y);
            m.z = fmaxf(fmaxf(col[j].z, col[j + 1].z), col[j + 2].z);
            m.w = fmaxf(fmaxf(col[j].w, col[j + 1].w), col[j + 2].w);
            out[(((int64_t)x * Y + y) * Z + z) * O4 + c] = m;          // O4: row stride of the (possibly wider) output
        }
    }
}

// r4 -- separable form through LDS: workgroup = brick of PB^3 output voxels x a slab of PCS float4 channels.  The (PB+2)^3 halo brick is
// read ONCE (2.4 loads per output instead of 27; out-of-grid voxels = -inf, the reference's padding), then max over z, over y, over x
// between two LDS buffers; max is exact and order-free, so the result equals the tap-by-tap kernel's bit for bit.
// 24x12x24x128 (the geometry2 pool): 32 bricks x 8 slabs = 256 workgroups, 64 KB of LDS.
constexpr int PB = 6, PH = PB + 2, PCS = 4;
__global__ __launch_bounds__(256) void maxpool3_lds_kernel(const float4 *__restrict__ in, int X, int Y, int Z, int C4,
                                                           float4 *__restrict__ out, int O4, int nbx, int nby, int nbz)
{
    extern __shared__ __attribute__((aligned(16))) float4 pl[];
    float4 *A = pl, *B = pl + PH * PH * PH * PCS;                // A: [x PH][y PH][z PH][PCS]; B: z pass [PH][PH][PB], later reused
    const float ninf = -__builtin_huge_valf();
    const int tid = threadIdx.x;
    // work list: slab-major inside a brick so that the workgroups of one brick sit on one XCD's L2 (block b runs on XCD b % 8)
    int wid;
    {
        const int nb = gridDim.x, xcd = blockIdx.x % 8, idx = blockIdx.x / 8, qd = nb / 8, rm = nb % 8;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    }
    const int nslab = C4 / PCS;
    const int slab = wid % nslab, brick = wid / nslab;
    const int bz = brick % nbz, by = (brick / nbz) % nby, bx = brick / (nbz * nby);
    const int x0 = bx * PB - 1, y0 = by * PB - 1, z0 = bz * PB - 1;
    // ---- halo brick -> A (every load requested before the first store)
    constexpr int NLD = PH * PH * PH * PCS / 256;                // 8
    float4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + i * 256, c = e % PCS, vox = e / PCS;
        const int hz = vox % PH, hy = (vox / PH) % PH, hx = vox / (PH * PH);
        const int gx = x0 + hx, gy = y0 + hy, gz = z0 + hz;
        const bool ok = (unsigned)gx < (unsigned)X && (unsigned)gy < (unsigned)Y && (unsigned)gz < (unsigned)Z;
        const int64_t off = ok ? (((int64_t)gx * Y + gy) * Z + gz) * C4 + slab * PCS + c : 0;
        const float4 t = in[off];
        v[i] = ok ? t : make_float4(ninf, ninf, ninf, ninf);
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) A[tid + i * 256] = v[i];
    __syncthreads();
    auto mx3 = [](const float4 &a, const float4 &b, const float4 &c) {
        return make_float4(fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z), fmaxf(fmaxf(a.w, b.w), c.w));
    };
    // ---- z pass: A [PH][PH][PH] -> B [PH][PH][PB]
    for (int e = tid; e < PH * PH * PB * PCS; e += 256) {
        const int c = e % PCS, r = e / PCS, z = r % PB, xy = r / PB;
        const float4 *p = A + (xy * PH + z) * PCS + c;
        B[e] = mx3(p[0], p[PCS], p[2 * PCS]);
    }
    __syncthreads();
    // ---- y pass: B [PH][PH][PB] -> A [PH][PB][PB]
    for (int e = tid; e < PH * PB * PB * PCS; e += 256) {
        const int c = e % PCS, r = e / PCS, z = r % PB, y = (r / PB) % PB, x = r / (PB * PB);
        const float4 *p = B + ((x * PH + y) * PB + z) * PCS + c;
        A[e] = mx3(p[0], p[PB * PCS], p[2 * PB * PCS]);
    }
    __syncthreads();
    // ---- x pass: A [PH][PB][PB] -> out
    for (int e = tid; e < PB * PB * PB * PCS; e += 256) {
        const int c = e % PCS, r = e / PCS, z = r % PB, y = (r / PB) % PB, x = r / (PB * PB);
        const int gx = x0 + 1 + x, gy = y0 + 1 + y, gz = z0 + 1 + z;
        if (gx < X && gy < Y && gz < Z) {
            const float4 *p = A + ((x * PB + y) * PB + z) * PCS + c;
            out[(((int64_t)gx * Y + gy) * Z + gz) * O4 + slab * PCS + c] = mx3(p[0], p[PB * PB * PCS], p[2 * PB * PB * PCS]);
        }
    }
}

// [rows][cols] -> [cols][rows] tiled transpose; the large dimension is mapped to gridDim.x
template <bool ROWS_ON_X>
__global__ void transpose_kernel(const float *__restrict__ src, int64_t rows, int64_t cols, float *__restrict__ dst)
{
    __shared__ float tile[32][33];
    const int64_t r0 = (int64_t)(ROWS_ON_X ? blockIdx.x : blockIdx.y) * 32;
    const int64_t c0 = (int64_t)(ROWS_ON_X ? blockIdx.y : blockIdx.x) * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int64_t r = r0 + j;
        const int64_t c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? src[r * cols + c] : 0.0f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int64_t c = c0 + j;
        const int64_t r = r0 + threadIdx.x;
        if (r < rows && c < cols) dst[c * rows + r] = tile[threadIdx.x][j];
    }
}

// raw signed distance field (file order: x fastest, then y, then z) -> the network's 2-channel input
// (lib/datasets/dataset.py:54-70): c0 = |clamp(v, -T, T)| (T - that with FLIP_TSDF, log of it with LOG_TSDF), c1 = v > -1.
// One (x,z) tile per block and y slice: reads coalesced along x, writes coalesced along z (the innermost spatial axis of the
// conv stack's channels-last layout), both channels of a voxel in one 8-byte store.
__global__ void tsdf_encode_kernel(const float *__restrict__ sdf, int X, int Y, int Z, float trunc, int mode,
                                   float *__restrict__ out, int64_t os_c, int64_t os_x, int64_t os_y, int64_t os_z)
{
    __shared__ float tile[32][33];
    const int y = blockIdx.z;
    const int x0 = blockIdx.x * 32, z0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int z = z0 + j, x = x0 + threadIdx.x;
        tile[j][threadIdx.x] = (z < Z && x < X) ? sdf[((int64_t)z * Y + y) * X + x] : 0.0f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int x = x0 + j, z = z0 + threadIdx.x;
        if (x >= X || z >= Z) continue;
        const float v = tile[threadIdx.x][j];
        float a = fabsf(fminf(fmaxf(v, -trunc), trunc));
        if (mode == 1) a = trunc - a;
        else if (mode == 2) a = logf(a);
        const float occ = v > -1.0f ? 1.0f : 0.0f;
        float *dst = out + x * os_x + y * os_y + z * os_z;
        if (os_c == 1) *reinterpret_cast<float2 *>(dst) = make_float2(a, occ);
        else { dst[0] = a; dst[os_c] = occ; }
    }
}

// grid-stride 16-byte copy; src may be pinned host memory mapped into the device's address space (the loads cross PCIe).
// FEW workgroups, MANY loads in flight per lane: a resident wave of ANY kernel keeps a Winograd workgroup (512 registers per lane = a
// SIMD's whole file, 148 KB of LDS) off its CU for as long as it lives, and a wave that waits on the link lives long -- the first
// version (64 workgroups, 4 loads in flight) cost four chunk pipelines 13-17 % of their throughput although the link was idle half of
// the time; 8 workgroups x 256 lanes x 8 x 16 B = 256 KB in flight cover the link's latency-bandwidth product several times over and
// occupy 8 of the 256 CUs.
typedef float upl4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void upload_kernel(const upl4 *__restrict__ src, upl4 *__restrict__ dst, int64_t n4)
{
    constexpr int U = 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        upl4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n4; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

// ---- host -> graph mailbox (include/sis3d.h): kernels inside a captured graph read what changes per chunk from a pinned host ring
struct MailSlot { unsigned long long src, dst; float origin[3]; unsigned flags; unsigned long long next_src, pad[3]; };
static_assert(sizeof(MailSlot) == 64, "mail slot layout");
constexpr int MAIL_SLOT_WORD = 8;          // state[8..23]: the fetched slot

// r6 -- slot integrity (the kernels behind the fetch dereference raw 64-bit pointers a host thread wrote).  The producer
// stamps every slot with its sequence number (pad[0] = slots written before it) and a check word over the pointers (pad[1]); the
// fetch -- the only kernel that reads the ring -- accepts the slot only if the number is the consumed-slot count (a STALE slot the
// host has not rewritten yet, or one it has LAPPED, carries another number) and the check word matches (a torn / half-written slot).
// A rejected slot is parked with its three pointers zeroed, so the upload, the piggyback row and the post touch nothing, and the
// reason goes to state[1] (sticky) -> progress[1] on the host, where ops.Mailbox raises.
__host__ __device__ inline unsigned long long mail_check_word(unsigned long long src, unsigned long long dst, unsigned long long next_src,
                                                              unsigned flags, unsigned long long seq)
{
    auto rotl = [](unsigned long long v, int r) { return (v << r) | (v >> (64 - r)); };
    return src ^ rotl(dst, 17) ^ rotl(next_src, 31) ^ ((unsigned long long)flags << 40) ^ (seq * 0x9E3779B97F4A7C15ull) ^ 0x5151D3D3ull;
}

// ONE read of the slot across the link (four 16-byte loads), parked in device memory for the launches behind it
__global__ __launch_bounds__(64) void mail_fetch_kernel(const upl4 *__restrict__ ring, int ring_size, unsigned *__restrict__ state)
{
    const unsigned k = state[0];
    if (threadIdx.x < 4) {
        const upl4 v = __builtin_nontemporal_load(ring + (size_t)(k % (unsigned)ring_size) * 4 + threadIdx.x);
        reinterpret_cast<upl4 *>(state + MAIL_SLOT_WORD)[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MailSlot *s = reinterpret_cast<MailSlot *>(state + MAIL_SLOT_WORD);
        unsigned err = 0;
        if ((unsigned)s->pad[0] != k || (s->pad[0] >> 32) != 0) err = 1;                                     // stale or lapped
        else if (s->pad[1] != mail_check_word(s->src, s->dst, s->next_src, s->flags, s->pad[0])) err = 2;   // torn
        if (err) {
            s->src = s->dst = s->next_src = 0;
            s->flags &= ~5u;                          // no origin, no staged copy
            if (state[1] == 0) state[1] = err;
        }
    }
}

constexpr int MAIL_STAGED_WORD = 24;       // state[24..25]: the source whose chunk the staging buffer holds (the piggyback row of conv3d_wino.hip)

__global__ __launch_bounds__(256) void mail_upload_kernel(const unsigned *__restrict__ state, upl4 *__restrict__ dst, int64_t n4,
                                                          float *__restrict__ origin_dst, const upl4 *__restrict__ staged)
{
    const MailSlot s = *reinterpret_cast<const MailSlot *>(state + MAIL_SLOT_WORD);      // device memory, uniform
    if (blockIdx.x == 0 && threadIdx.x < 3 && origin_dst && (s.flags & 1u)) origin_dst[threadIdx.x] = s.origin[threadIdx.x];
    const upl4 *src = reinterpret_cast<const upl4 *>(s.src);
    // the previous pass already pulled this chunk into the staging buffer (slot flag bit 2: the host announced it as that pass's
    // next_src and has not touched it since): copy it from there at HBM speed
    const bool from_stage = staged && src && (s.flags & 4u) &&
                            *reinterpret_cast<const unsigned long long *>(state + MAIL_STAGED_WORD) == s.src;
    if (from_stage) src = staged;
    // a HOST source is pulled across the link by the first 8 workgroups only (a wave that waits on the link must not sit on many CUs:
    // see upload_kernel); a DEVICE source (flags bit 1, or the staged copy) is copied by the whole grid at HBM speed
    const int active = ((s.flags & 2u) || from_stage) ? (int)gridDim.x : (gridDim.x < 8 ? (int)gridDim.x : 8);
    if (!src || (int)blockIdx.x >= active) return;
    constexpr int U = 8;
    const int64_t stride = (int64_t)active * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        upl4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n4; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

__global__ __launch_bounds__(256) void mail_post_kernel(unsigned *__restrict__ state, const float *__restrict__ block_src, int64_t n,
                                                        unsigned long long *__restrict__ progress)
{
    const MailSlot s = *reinterpret_cast<const MailSlot *>(state + MAIL_SLOT_WORD);
    float *dst = reinterpret_cast<float *>(s.dst);
    if (dst && block_src)
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = block_src[i];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned k = state[0] + 1u;
        state[0] = k;
        if (progress) {
            if (state[1]) __hip_atomic_store(progress + 1, (unsigned long long)state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(progress, (unsigned long long)k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

} // namespace

extern "C" int sis3d_mail_fetch(const void *ring, int ring_size, uint32_t *state, sis3d_stream_t stream)
{
    if (!ring || ring_size <= 0 || !state || ((uintptr_t)ring & 15) || ((uintptr_t)state & 15)) return SIS3D_EINVAL;
    hipLaunchKernelGGL(mail_fetch_kernel, dim3(1), dim3(64), 0, as_stream(stream), (const upl4 *)ring, ring_size, (unsigned *)state);
    return sis3d_check_launch();
}

extern "C" int sis3d_mail_upload(const uint32_t *state, float *input_dst, int64_t n, float *origin_dst, const float *staged, int workgroups,
                                 sis3d_stream_t stream)
{
    if (!state || !input_dst || n <= 0 || (n & 3) || ((uintptr_t)input_dst & 15) || ((uintptr_t)staged & 15) || workgroups < 0)
        return SIS3D_EINVAL;
    static const int env_wg = [] { const char *e = getenv("SIS3D_MAIL_WGS"); return e ? atoi(e) : 0; }();      // tuning hook
    const int wg = workgroups > 0 ? workgroups : (env_wg > 0 ? env_wg : 64);
    hipLaunchKernelGGL(mail_upload_kernel, dim3(wg), dim3(256), 0, as_stream(stream), (const unsigned *)state, (upl4 *)input_dst, n / 4,
                       origin_dst, (const upl4 *)staged);
    return sis3d_check_launch();
}

extern "C" int sis3d_mail_post(uint32_t *state, const float *block_src, int64_t n, uint64_t *progress, sis3d_stream_t stream)
{
    if (!state || n < 0 || (n > 0 && !block_src)) return SIS3D_EINVAL;
    hipLaunchKernelGGL(mail_post_kernel, dim3(1), dim3(256), 0, as_stream(stream), (unsigned *)state, block_src, n,
                       (unsigned long long *)progress);
    return sis3d_check_launch();
}

extern "C" int sis3d_upload_f32(const float *src, float *dst, int64_t n, int workgroups, sis3d_stream_t stream)
{
    if (!src || !dst || n <= 0 || (n & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15) || workgroups < 0) return SIS3D_EINVAL;
    static const int env_wg = [] { const char *e = getenv("SIS3D_UPLOAD_WGS"); return e ? atoi(e) : 0; }();      // tuning hook
    const int wg = workgroups > 0 ? workgroups : (env_wg > 0 ? env_wg : 8);
    hipLaunchKernelGGL(upload_kernel, dim3(wg), dim3(256), 0, as_stream(stream), (const upl4 *)src, (upl4 *)dst, n / 4);
    return sis3d_check_launch();
}

extern "C" int sis3d_tsdf_encode(const float *sdf, int X, int Y, int Z, int Yout, float truncated, int mode, float *out,
                                 int64_t os_c, int64_t os_x, int64_t os_y, int64_t os_z, sis3d_stream_t stream)
{
    if (!sdf || !out || X <= 0 || Y <= 0 || Z <= 0 || Yout <= 0 || Yout > Y || mode < 0 || mode > 2) return SIS3D_EINVAL;
    if (Yout > 65535 || cdiv(Z, 32) > 65535) return SIS3D_EUNSUPPORTED;
    if (os_c == 1 && ((os_x | os_y | os_z) & 1)) return SIS3D_EINVAL;      // 8-byte stores need even strides
    hipLaunchKernelGGL(tsdf_encode_kernel, dim3(cdiv(X, 32), cdiv(Z, 32), Yout), dim3(32, 8), 0, as_stream(stream), sdf, X, Y, Z,
                       truncated, mode, out, os_c, os_x, os_y, os_z);
    return sis3d_check_launch();
}

extern "C" int sis3d_maxpool3d_3x3x3(const float *in, int X, int Y, int Z, int C, float *out, int out_stride, int out_coff,
                                     sis3d_stream_t stream)
{
    if (!in || !out || X <= 0 || Y <= 0 || Z <= 0 || C <= 0 || (C % 4)) return SIS3D_EINVAL;
    if (out_stride < C || (out_stride % 4) || out_coff < 0 || (out_coff % 4) || out_coff + C > out_stride) return SIS3D_EINVAL;
    auto threads = [&](int zseg) { return (int64_t)X * Y * ((Z + zseg - 1) / zseg) * (C / 4); };
    auto go = [&](auto kern, int zseg) {
        const int64_t total = threads(zseg);
        const int blocks = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, as_stream(stream), (const float4 *)in, X, Y, Z, C / 4,
                           (float4 *)(out + out_coff), out_stride / 4);
        return sis3d_check_launch();
    };
    static const int force = [] { const char *e = getenv("SIS3D_POOL_ZSEG"); return e ? atoi(e) : 0; }();   // tuning hook
    // r4: the separable LDS form wherever the slab split is exact and the grid gives it enough workgroups (SIS3D_POOL_LDS=0: off)
    static const int use_lds = [] { const char *e = getenv("SIS3D_POOL_LDS"); return e ? atoi(e) : 1; }();
    if (use_lds && force == 0 && (C / 4) % PCS == 0) {
        const int nbx = cdiv(X, PB), nby = cdiv(Y, PB), nbz = cdiv(Z, PB);
        const int64_t nwg = (int64_t)nbx * nby * nbz * ((C / 4) / PCS);
        if (nwg >= 200 && nwg <= 0x7fffffff) {       // fewer workgroups: the tap-by-tap kernel spreads better (24x12x24x64: 4.5 vs 4.8 us)
            constexpr int lds = (PH * PH * PH + PH * PH * PB) * PCS * (int)sizeof(float4);      // 57,344 B
            hipLaunchKernelGGL(maxpool3_lds_kernel, dim3((unsigned)nwg), dim3(256), lds, as_stream(stream), (const float4 *)in, X, Y, Z, C / 4,
                               (float4 *)(out + out_coff), out_stride / 4, nbx, nby, nbz);
            return sis3d_check_launch();
        }
    }
    if (force == 8) return go(maxpool3_kernel<8>, 8);
    if (force == 4) return go(maxpool3_kernel<4>, 4);
    if (force == 2) return go(maxpool3_kernel<2>, 2);
    if (force == 1) return go(maxpool3_kernel<1>, 1);
    // measured with the tap-by-tap form (tools/pool_time.py): 48x24x48x64 map 27.6 / 24.4 / 25.3 / 33.6 us for ZSEG 1/2/4/8,
    // 24x12x24x128 map 10.2 / 9.8 / 11.6 / 14.6 us, 24x12x24x64 map 7.6 / 8.9 / 11.0 / 14.6 us; with batched tap loads the
    // chosen variants run in 21.5 / 6.8 / 4.3 us
    if (threads(2) >= 200000) return go(maxpool3_kernel<2>, 2);
    return go(maxpool3_kernel<1>, 1);
}

// planar (C, nvox) -> channels-last (nvox, C): a transpose of a [C][nvox] matrix
extern "C" int sis3d_planar_to_cl(const float *in, int C, int64_t nvox, float *out, sis3d_stream_t stream)
{
    if (!in || !out || C <= 0 || nvox <= 0) return SIS3D_EINVAL;
    hipLaunchKernelGGL((transpose_kernel<false>), dim3(cdiv(nvox, 32), cdiv(C, 32)), dim3(32, 8), 0, as_stream(stream), in,
                       (int64_t)C, nvox, out);
    return sis3d_check_launch();
}

extern "C" int sis3d_cl_to_planar(const float *in, int C, int64_t nvox, float *out, sis3d_stream_t stream)
{
    if (!in || !out || C <= 0 || nvox <= 0) return SIS3D_EINVAL;
    hipLaunchKernelGGL((transpose_kernel<true>), dim3(cdiv(nvox, 32), cdiv(C, 32)), dim3(32, 8), 0, as_stream(stream), in, nvox,
                       (int64_t)C, out);
    return sis3d_check_launch();
}
