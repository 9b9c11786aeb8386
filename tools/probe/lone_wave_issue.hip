// What does ONE wave on an otherwise idle CU pay per instruction?  (k_ed_detect's smart-routing walk is one wave per frame.)
// hipcc --offload-arch=gfx950 -O3 lone_wave_issue.hip -o /tmp/lone_wave_issue && /tmp/lone_wave_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
__global__ void probe(long long* out, int n)
{
    long long t0, t1;
    int a = n, v = threadIdx.x;
    // 1: dependent SALU chain
    t0 = __builtin_readcyclecounter();
    asm volatile(REP64("s_add_i32 %0, %0, 1\n") : "+s"(a));
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    // 2: dependent VALU chain
    t0 = __builtin_readcyclecounter();
    asm volatile(REP64("v_add_u32 %0, %0, 1\n") : "+v"(v));
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    // 3: readlane -> salu -> readlane (lane select depends on the previous result)
    int l = a & 63;
    t0 = __builtin_readcyclecounter();
    asm volatile(REP64("v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63\n") : "+s"(l) : "v"(v));
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    // 4: a loop of 64 taken branches (s_add, s_cmp, s_cbranch)
    int c = 0;
    t0 = __builtin_readcyclecounter();
    asm volatile("1:\n s_add_i32 %0, %0, 1\n s_cmp_lt_i32 %0, 64\n s_cbranch_scc1 1b\n" : "+s"(c));
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[3] = t1 - t0;
    // 5: VALU compare -> vcc -> s_and vcc -> branch (the compiler's uniform-branch idiom), 64 times not taken
    t0 = __builtin_readcyclecounter();
    asm volatile(REP64("v_cmp_eq_u32 vcc, %0, %1\n s_and_b64 vcc, exec, vcc\n s_cbranch_vccz 2f\n 2:\n") :: "v"(v), "s"(a) : "vcc");
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[4] = t1 - t0;
    // 6: SALU -> VALU (sgpr operand) -> readfirstlane -> SALU ping-pong
    int s = a;
    t0 = __builtin_readcyclecounter();
    asm volatile(REP64("v_add_u32 %1, %0, %1\n v_readfirstlane_b32 %0, %1\n s_add_i32 %0, %0, 1\n") : "+s"(s), "+v"(v));
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[5] = t1 - t0;
    // 7: 64 taken forward branches (s_branch over one instruction)
    t0 = __builtin_readcyclecounter();
    asm volatile(REP64("s_branch 3f\n s_nop 0\n 3:\n s_add_i32 %0, %0, 1\n") : "+s"(a));
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[6] = t1 - t0;
    if (threadIdx.x == 0) out[7] = a + v + l + c + s;
}
int main()
{
    long long* d; hipMalloc(&d, 64);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1); hipDeviceSynchronize(); }
    long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    const char* name[7] = { "dependent s_add", "dependent v_add", "readlane + s_and (dependent lane select)", "loop: s_add, s_cmp, taken s_cbranch", "v_cmp, s_and vcc, s_cbranch (not taken)", "v_add(sgpr), v_readfirstlane, s_add", "taken s_branch + s_add" };
    for (int i = 0; i < 7; ++i) printf("%-45s %6.1f cycles per repetition (64 repetitions, %lld total)\n", name[i], h[i] / 64.0, h[i]);
    return 0;
}
