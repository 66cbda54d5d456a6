// Latency / issue-rate probe for the instruction kinds the FPS iteration is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/diag/lat_probe.hip -o tools/diag/lat_probe ; run on the GPU box.
// Prints ticks of s_memtime per operation and the s_memtime frequency against s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP 4096

__device__ __forceinline__ unsigned long long mt() {
    unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    return t;
}

template <int KIND>
__global__ void probe(unsigned long long *out, float seed, int nwaves_barrier) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    float a = seed + lane, b = seed * 0.5f, c = 1.0f, d = 2.0f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f pa = {a, b}, pb = {c, d};
    int iv = lane;
    lds[threadIdx.x] = a;
    __syncthreads();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = mt();
    if (KIND == 0) {  // dependent v_fma_f32 chain
#pragma unroll 64
        for (int i = 0; i < REP; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    } else if (KIND == 1) {  // 4 independent chains (issue rate)
#pragma unroll 16
        for (int i = 0; i < REP / 4; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(pa.x) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(pa.y) : "v"(b), "v"(c));
        }
    } else if (KIND == 2) {  // dependent v_pk_fma_f32 chain
#pragma unroll 64
        for (int i = 0; i < REP; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pa) : "v"(pb));
    } else if (KIND == 3) {  // dependent DPP max chain with the required nops
#pragma unroll 64
        for (int i = 0; i < REP; ++i)
            asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(iv));
    } else if (KIND == 4) {  // VALU -> readlane -> SALU -> VALU round trip
#pragma unroll 64
        for (int i = 0; i < REP; ++i) {
            int s;
            asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(iv));
            asm volatile("v_add_u32 %0, %1, %0" : "+v"(iv) : "s"(s));
        }
    } else if (KIND == 5) {  // LDS read round trip (dependent address)
        int addr = (lane * 4) & 4095;
#pragma unroll 16
        for (int i = 0; i < REP; ++i) {
            float v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            addr = (addr + (__builtin_bit_cast(int, v) & 4)) & 4095;
        }
        iv = addr;
    } else if (KIND == 6) {  // barrier round trip, all waves arrive together
        for (int i = 0; i < REP; ++i) __builtin_amdgcn_s_barrier();
    } else if (KIND == 7) {  // v_cmp -> v_cndmask (SGPR mask) -> v_max chain: the arg-max tracking pattern
#pragma unroll 32
        for (int i = 0; i < REP; ++i) {
            asm volatile("v_cmp_gt_f32_e64 s[4:5], %1, %0\n\tv_max_f32 %0, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %2, %2, 7, s[4:5]"
                         : "+v"(a), "+v"(b), "+v"(iv) : : "s4", "s5");
        }
    } else if (KIND == 8) {  // LDS write + barrier + LDS read (the exchange)
        for (int i = 0; i < REP; ++i) {
            if (lane == 0) lds[(threadIdx.x >> 6) * 8 + (i & 1) * 512] = a;
            __syncthreads();
            a += lds[(lane & 15) * 8 + (i & 1) * 512];
        }
    } else if (KIND == 9) {  // s_set_gpr_idx + v_mov + readlane (dynamic register fetch)
        int s = 2;
#pragma unroll 32
        for (int i = 0; i < REP; ++i) {
            asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, %0\n\ts_set_gpr_idx_off\n\tv_readlane_b32 %1, %0, 1\n\ts_and_b32 %1, %1, 0"
                         : "+v"(iv), "+s"(s));
        }
    }
    const unsigned long long t1 = mt();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = rt1 - rt0; }
    if (a + b + c + d + pa.x + pa.y + iv == 12345.678f) out[2] = 1;
}

template <int KIND>
static void run(const char *name, int threads, unsigned long long *dout) {
    unsigned long long h[3];
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(threads), 0, 0, dout, 1.0f, threads / 64);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-52s threads %4d: %7.2f ticks/op   (memtime %.0f MHz)\n", name, threads, (double)h[0] / REP,
           (double)h[0] / ((double)h[1] / 100.0));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long *dout;
    hipMalloc(&dout, 64);
    for (int threads : {64, 256, 512, 1024}) {
        run<0>("dependent v_fma_f32", threads, dout);
        run<1>("independent v_fma_f32 (per instruction)", threads, dout);
        run<2>("dependent v_pk_fma_f32", threads, dout);
        run<3>("dependent s_nop 1 + v_max_i32_dpp", threads, dout);
        run<4>("v_readlane -> SGPR -> v_add", threads, dout);
        run<5>("ds_read_b32 dependent round trip", threads, dout);
        run<6>("s_barrier", threads, dout);
        run<7>("v_cmp + v_max + nop + v_cndmask (3 VALU)", threads, dout);
        run<8>("lds write + barrier + lds read", threads, dout);
        // run<9> (s_set_gpr_idx + v_mov + readlane) hangs on gfx950 as written: left out of the run
    }
    return 0;
}
