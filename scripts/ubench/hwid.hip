// Where do the waves of a 256-thread workgroup land?  (the four-wave pipelined kernel gives wave w of
// workgroup b the stage (w + rot(b)) % 4.  If a SIMD hosts the SAME stage of every workgroup on its CU, the CU is paced by
// the heaviest stage x 4 and the SIMDs with light stages idle; if it hosts one wave of each stage, by the sum.)
// Launches the headline shape -- 1024 workgroups x 256 threads, 24 640 B of LDS each -- with enough work that all are
// resident together, reads HW_REG_HW_ID / HW_REG_XCC_ID in every wave, and prints, per rotation rule the generator offers
// (OGC_ROT 0..4) and for the hardware rule (simd_id + wave slot), how many SIMDs end up with 4 distinct stages.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid scripts/ubench/hwid.hip && /tmp/hwid
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void __launch_bounds__(256) probe(unsigned* out, int spin)
{
    __shared__ float pad[24640 / 4];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); // HW_REG_XCC_ID
    float acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
    pad[threadIdx.x] = acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = hw;
        out[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = xcc + (pad[(threadIdx.x + 1) & 255] == 12345.0f);
    }
}

int main()
{
    const int nwg = 1024;
    unsigned* d;
    hipMalloc(&d, nwg * 4 * 2 * 4);
    probe<<<nwg, 256>>>(d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nwg * 4 * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    auto field = [](unsigned v, int lo, int n) { return (v >> lo) & ((1u << n) - 1u); };
    // gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    printf("first workgroups: block -> (xcc, se, sh, cu, simd, slot) of waves 0..3\n");
    for (int b = 0; b < 24; ++b) {
        printf("  block %4d:", b);
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1] & 15u;
            printf("  (%u,%u,%u,%2u,%u,%u)", x, field(hw, 13, 3), field(hw, 12, 1), field(hw, 8, 4), field(hw, 4, 2), field(hw, 0, 4));
        }
        printf("\n");
    }
    // per SIMD: which stages would it host under each rule?
    const char* names[6] = {"rot 0 (wave index)", "rot 1 (+ blockIdx)", "rot 2 (+ blockIdx >> 3)", "rot 3 (+ blockIdx >> 5)", "rot 4 (+ blockIdx >> 8)",
                            "hardware (simd_id + wave slot)"};
    for (int rule = 0; rule < 6; ++rule) {
        std::map<unsigned long long, std::vector<int>> simd_stages;
        int bad_wg = 0;
        for (int b = 0; b < nwg; ++b) {
            std::set<int> st;
            for (int w = 0; w < 4; ++w) {
                const unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1] & 15u;
                const unsigned rot[5] = {0u, (unsigned)b, (unsigned)b >> 3, (unsigned)b >> 5, (unsigned)b >> 8};
                const int stage = rule < 5 ? (w + rot[rule]) % 4 : (field(hw, 4, 2) + field(hw, 0, 4)) % 4;
                st.insert(stage);
                const unsigned long long key = ((unsigned long long)x << 32) | (hw & 0xFFF0u); // xcc, se, sh, cu, pipe, simd
                simd_stages[key].push_back(stage);
            }
            if (st.size() != 4) ++bad_wg;
        }
        int hist[5] = {0, 0, 0, 0, 0}, waves_hist[9] = {0};
        for (auto& kv : simd_stages) {
            std::set<int> s(kv.second.begin(), kv.second.end());
            hist[s.size()]++;
            waves_hist[kv.second.size() > 8 ? 8 : kv.second.size()]++;
        }
        printf("%-32s SIMDs used %zu; SIMDs hosting 1/2/3/4 distinct stages: %d %d %d %d; workgroups whose 4 waves do not get 4 distinct stages: %d\n", names[rule],
               simd_stages.size(), hist[1], hist[2], hist[3], hist[4], bad_wg);
        if (rule == 0) printf("%-32s waves per SIMD histogram (1..8): %d %d %d %d %d %d %d %d\n", "", waves_hist[1], waves_hist[2], waves_hist[3], waves_hist[4],
                              waves_hist[5], waves_hist[6], waves_hist[7], waves_hist[8]);
    }
    return 0;
}
