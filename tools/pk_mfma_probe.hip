// Probe (not part of the product): do packed fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) give the same results while another
// stream of the SAME process keeps the matrix cores busy?  Stream A runs a register-only v_mfma_f32_32x32x16_f16 loop; stream B repeats a short
// kernel whose threads push one value pair through a chain of packed (MODE 0) or scalar (MODE 1) multiply-adds; every launch is compared bit for bit with the
// result computed while stream A was idle, and the lane positions of the differing threads are counted.
//   hipcc --offload-arch=gfx950 -O3 tools/pk_mfma_probe.hip -o tools/build/pk_mfma_probe;  tools/build/pk_mfma_probe [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_hog(int iters, float *sink)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x16 c0 = {}, c1 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
    if (c0[0] + c1[3] == 12345.f) sink[threadIdx.x] = c0[1];
}

template <int MODE>
__global__ __launch_bounds__(256) void chain(const float *__restrict__ in, int n, int reps, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = in[2 * i], y = in[2 * i + 1];
    const float m = 0.999f, a = 0.0007f;
    if (MODE == 0) {
        f32x2v v = {x, y};
        const f32x2v mm = {m, m}, aa = {a, a};
        for (int r = 0; r < reps; ++r) { v = v * mm; v = v + aa; v = __builtin_elementwise_fma(v, mm, aa); }
        x = v.x; y = v.y;
    } else if (MODE == 4) {                                                        // packed, with the operand halves swapped / broadcast (op_sel / op_sel_hi modifiers)
        f32x2v v = {x, y};
        const f32x2v mm = {m, 1.001f}, aa = {a, -a};
        for (int r = 0; r < reps; ++r) {
            v = __builtin_shufflevector(v, v, 1, 0) * mm;                           // lo = v.hi * mm.lo, hi = v.lo * mm.hi
            v = v + __builtin_shufflevector(aa, aa, 1, 0);
            v = __builtin_shufflevector(v, v, 0, 0) * mm + __builtin_shufflevector(v, v, 1, 1) * aa;   // broadcasts
        }
        x = v.x; y = v.y;
    } else {
        for (int r = 0; r < reps; ++r) {
            x = __fmul_rn(x, m); y = __fmul_rn(y, m); x = __fadd_rn(x, a); y = __fadd_rn(y, a); x = fmaf(x, m, a); y = fmaf(y, m, a);
            asm volatile("" : "+v"(x), "+v"(y));                                   // (keeps the scalar form scalar)
        }
    }
    out[2 * i] = x; out[2 * i + 1] = y;
}

// MODE 2 / 3: the operands of the packed (2) / scalar (3) arithmetic are gathered in the same iteration (featmap_kernel's pattern: four loads, then products of pairs)
template <int MODE>
__global__ __launch_bounds__(256) void gather_chain(const float *__restrict__ in, int n, int reps, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float hx = 0.25f + 0.5f * (float)(i & 7) * 0.125f, lx = 1.f - hx, hy = 0.375f, ly = 0.625f;
    float o[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        const float *p = in + (size_t)c * 4801;
        const int b = (i >> 3) % 4700;
        const float p00 = p[b], p01 = p[b + 1], p10 = p[b + 80], p11 = p[b + 81];
        if (MODE == 2) {
            const f32x2v t = f32x2v{p00, p10} * f32x2v{hx, hx} + f32x2v{p01, p11} * f32x2v{lx, lx};
            o[c] = hy * t.x + ly * t.y;
        } else {
            o[c] = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
        }
    }
    float4 *dst = reinterpret_cast<float4 *>(out + (size_t)i * 12);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]); dst[1] = make_float4(o[4], o[5], o[6], o[7]); dst[2] = make_float4(o[8], o[9], o[10], o[11]);
}

template <int MODE>
static void run_gather(const char *name, int launches, const float *din, float *dout, int n, hipStream_t sa, hipStream_t sb, float *sink, bool own_hog)
{
    std::vector<unsigned> ref((size_t)12 * n), cur((size_t)12 * n);
    gather_chain<MODE><<<(n + 255) / 256, 256, 0, sb>>>(din, n, 1, dout);
    (void)hipStreamSynchronize(sb);
    (void)hipMemcpy(ref.data(), dout, (size_t)n * 48, hipMemcpyDeviceToHost);
    int bad = 0; long q[4] = {0, 0, 0, 0};
    for (int r = 0; r < launches; ++r) {
        if (own_hog && (r & 15) == 0) mfma_hog<<<2048, 256, 0, sa>>>(60000, sink);
        (void)hipMemsetAsync(dout, 0xff, (size_t)n * 48, sb);
        gather_chain<MODE><<<(n + 255) / 256, 256, 0, sb>>>(din, n, 1, dout);
        (void)hipStreamSynchronize(sb);
        (void)hipMemcpy(cur.data(), dout, (size_t)n * 48, hipMemcpyDeviceToHost);
        int b = 0;
        for (int t = 0; t < n; ++t) { bool d = false; for (int c = 0; c < 12; ++c) d |= cur[(size_t)12 * t + c] != ref[(size_t)12 * t + c]; if (d) { ++b; ++q[(t & 63) >> 4]; } }
        if (b) ++bad;
    }
    (void)hipDeviceSynchronize();
    printf("%s: %d of %d launches differ; differing threads by lane quarter [0-15, 16-31, 32-47, 48-63]: %ld %ld %ld %ld\n", name, bad, launches, q[0], q[1], q[2], q[3]);
}

template <int MODE>
static void run(const char *name, int launches, const float *din, float *dout, int n, hipStream_t sa, hipStream_t sb, float *sink, bool own_hog)
{
    std::vector<unsigned> ref(2 * n), cur(2 * n);
    chain<MODE><<<(n + 255) / 256, 256, 0, sb>>>(din, n, 64, dout);
    (void)hipStreamSynchronize(sb);
    (void)hipMemcpy(ref.data(), dout, (size_t)n * 8, hipMemcpyDeviceToHost);
    int bad = 0; long q[4] = {0, 0, 0, 0};
    for (int r = 0; r < launches; ++r) {
        if (own_hog && (r & 15) == 0) mfma_hog<<<2048, 256, 0, sa>>>(60000, sink);           // ~ tens of ms of matrix work on every CU
        (void)hipMemsetAsync(dout, 0xff, (size_t)n * 8, sb);
        chain<MODE><<<(n + 255) / 256, 256, 0, sb>>>(din, n, 64, dout);
        (void)hipStreamSynchronize(sb);
        (void)hipMemcpy(cur.data(), dout, (size_t)n * 8, hipMemcpyDeviceToHost);
        int b = 0;
        for (int t = 0; t < n; ++t) if (cur[2 * t] != ref[2 * t] || cur[2 * t + 1] != ref[2 * t + 1]) { ++b; ++q[(t & 63) >> 4]; }
        if (b) ++bad;
    }
    (void)hipDeviceSynchronize();
    printf("%s: %d of %d launches differ from the launch with idle matrix cores; differing threads by lane quarter [0-15, 16-31, 32-47, 48-63]: %ld %ld %ld %ld\n", name, bad, launches,
           q[0], q[1], q[2], q[3]);
}

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 2000, n = 1 << 20;
    const bool own_hog = !(argc > 2 && atoi(argv[2]) == 0);                      // second argument 0: no matrix work of our own (another PROCESS provides the load)
    std::vector<float> h(2 * n);
    unsigned s = 777u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f) + 0.25f; }
    float *din, *dout, *sink;
    (void)hipMalloc(&din, (size_t)n * 8); (void)hipMalloc(&dout, (size_t)n * 8); (void)hipMalloc(&sink, 4096);
    (void)hipMemcpy(din, h.data(), (size_t)n * 8, hipMemcpyHostToDevice);
    hipStream_t sa, sb; (void)hipStreamCreate(&sa); (void)hipStreamCreate(&sb);
    run<0>("packed fp32 (v_pk_mul/add/fma_f32)", launches, din, dout, n, sa, sb, sink, own_hog);
    run<4>("packed fp32 with op_sel modifiers  ", launches, din, dout, n, sa, sb, sink, own_hog);
    run<1>("scalar fp32 (v_mul/add/fma_f32)   ", launches, din, dout, n, sa, sb, sink, own_hog);
    {   // gathered operands: n2 threads x 12 outputs; the planes are the first 12 x 4801 floats of din
        const int n2 = 480 * 640;
        float *dout2; (void)hipMalloc(&dout2, (size_t)n2 * 48);
        run_gather<2>("gathers + packed fp32 ", launches, din, dout2, n2, sa, sb, sink, own_hog);
        run_gather<3>("gathers + scalar fp32 ", launches, din, dout2, n2, sa, sb, sink, own_hog);
    }
    return 0;
}
