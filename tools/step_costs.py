"""VALU instructions of one node step / one triangle record, for the bench line's VALU roofline (benchlegs/ao.py) -- no GPU needed.
Two sources, both written to profiles/step_costs.json:
  * `loop_body`: the counts DESIGN 3 quotes, taken from the disassembly of the production kernel's loop (addresses of the LDS column and of
    the node array hoisted out of the loop): 136 for the ranked 4-wide step, 109 for the any-hit step that does not rank, 75 for a
    triangle record through the fp32 filter;
  * `probe`: this script compiles lh_kernels.hip with three one-step kernels appended (the step between a load and a store of the lane's
    state) and counts the v_* lines of each: an upper bound -- it includes the ~10-20 address computations a standalone step needs.
The bench prices an any-hit frame's node steps with `node_step4_unsorted` (it used the ranked step's 136).
  python tools/step_costs.py"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "%s/lucille_amd/csrc/lh_kernels.hip"
namespace {
template <int WHAT>
__global__ __launch_bounds__(256) void k_probe(lh_dev_scene_t sc, Lane *lanes, int *pends, float4 *tri)
{
    extern __shared__ int lds[];
    int (*stk)[LH_BLOCK] = (int (*)[LH_BLOCK])lds;
    Lane L = lanes[threadIdx.x]; int pend = pends[threadIdx.x]; uint32_t cn = 0;
    if (WHAT == 1) node_step4<false, LH_BLOCK, false, true>(L, pend, sc, stk, threadIdx.x, cn, 0, nullptr, 0u);
    if (WHAT == 2) node_step4<false, LH_BLOCK, false, false>(L, pend, sc, stk, threadIdx.x, cn, 0, nullptr, 0u);
    if (WHAT == 3) {
        const float4 ta = tri[3 * threadIdx.x], tb_ = tri[3 * threadIdx.x + 1], tc = tri[3 * threadIdx.x + 2];
        float t_hi;
        const int cls = lh_tri_filter(&L.r, ta.x, ta.y, ta.z, ta.w, tb_.x, tb_.y, tb_.z, tb_.w, tc.x, tc.z, tc.w, L.tb, &t_hi);
        L.cur = cls; L.tb = t_hi;
    }
    lanes[threadIdx.x] = L; pends[threadIdx.x] = pend;
}
template __global__ void k_probe<1>(lh_dev_scene_t, Lane *, int *, float4 *);
template __global__ void k_probe<2>(lh_dev_scene_t, Lane *, int *, float4 *);
template __global__ void k_probe<3>(lh_dev_scene_t, Lane *, int *, float4 *);
}
''' % ROOT
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "probe.hip"), "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "lucille_amd", "csrc"),
                           "-S", "--cuda-device-only", os.path.join(d, "probe.hip"), "-o", os.path.join(d, "probe.s")], stderr=subprocess.DEVNULL)
    lines = open(os.path.join(d, "probe.s")).read().split("\n")
probe = {}
i = 0
while i < len(lines):
    m = re.match(r"^_ZN\S*k_probeILi(\d)\S*:", lines[i])
    if m:
        j = i + 1; v = 0
        while "s_endpgm" not in lines[j]:
            v += 1 if re.match(r"v_", lines[j].strip()) else 0; j += 1
        probe[{"1": "node_step4_sorted", "2": "node_step4_unsorted", "3": "tri_filter"}[m.group(1)]] = v
        i = j
    i += 1
out = {"node_step4_sorted": 136, "node_step4_unsorted": 109, "tri_filter": 75,
       "loop_body_source": "DESIGN.md 3: the production kernel's loop body in the disassembly of lh_kernels.hip (r05)",
       "probe": probe, "probe_note": "standalone one-step kernels, v_* lines between the lane state's load and store: includes the address arithmetic the production loop hoists"}
json.dump(out, open(os.path.join(ROOT, "profiles", "step_costs.json"), "w"), indent=1)
print(json.dumps(out))
