"""bench.py: constants of the roofline formulas, HIP events on an explicit stream, ray upload, the gather ceiling, host facts."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
B_IN, B_OUT, B_TRI = 32, 16, 40   # SURVEY.md 8(d) algorithmic bytes per ray / per triangle test
B_NODE_SURVEY = 64                # a node visit is priced at 64 B whatever the record
def _step_costs():
    """VALU instructions of one 4-wide node step (ranked: closest hit; unranked: any hit) / one triangle record through the fp32 filter:
    profiles/step_costs.json (tools/step_costs.py; lh_walk.h, lh_filter.h: counted in the disassembly)"""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "step_costs.json")))
        return int(j["node_step4_sorted"]), int(j["node_step4_unsorted"]), int(j["tri_filter"])
    except (OSError, ValueError, KeyError):
        return 136, 109, 75


VALU_NODE_STEP, VALU_NODE_STEP_ANYHIT, VALU_TRI_STEP = _step_costs()
VALU_PEAK_TLANEOPS = 256 * 64 * 2.4e9 / 1e12   # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane operations per second (one wave64 VALU instruction per SIMD every 4 cycles)
B_NODE = {"f32": 64, "q16": 32, "q16x4": 64}   # per node visit: SURVEY's 64-B fp32 2-wide node, 32-B 16-bit grid 2-wide, 64-B 16-bit grid 4-wide
# check values of the canonical S-soup-1M dump on the UNMODIFIED reference
SOUP1M_CHECK = {1_000_000: (821_596, 87998.6606), 2_000_000: (1_644_156, 176110.93)}


def hip_events():
    """HIP events on an explicit stream, straight from libamdhip64 (torch.cuda.Event only
    sees torch's current stream; the kernel is launched on the stream we pass)."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipEventSynchronize.argtypes = [C.c_void_p]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    return hip


class EventPairs:
    def __init__(self, hip, n):
        self.hip = hip
        self.ev = [(C.c_void_p(), C.c_void_p()) for _ in range(n)]
        for a, b in self.ev:
            hip.hipEventCreate(C.byref(a)); hip.hipEventCreate(C.byref(b))
        self.k = 0

    def begin(self, sptr):
        self.hip.hipEventRecord(self.ev[self.k][0], sptr)

    def end(self, sptr):
        self.hip.hipEventRecord(self.ev[self.k][1], sptr); self.k += 1

    def ms(self):
        out = []
        for a, b in self.ev[:self.k]:
            v = C.c_float(); self.hip.hipEventElapsedTime(C.byref(v), a, b); out.append(v.value)
        return out


def upload_rays(scenes, torch, dev, state, n, keep_first=0):
    """n rays of the stream starting at `state` -> HBM (generated on the host in 10 M-ray pieces)"""
    d_org = torch.empty((n, 3), dtype=torch.float64, device=dev)
    d_dir = torch.empty((n, 3), dtype=torch.float64, device=dev)
    chunk = 10_000_000
    ho = np.empty((min(chunk, max(n, 1)), 3)); hd = np.empty((min(chunk, max(n, 1)), 3))
    first = None
    for b in range(0, n, chunk):
        m = min(chunk, n - b)
        _, _, state = scenes.soup_rays(m, state, ho, hd)
        d_org[b:b + m].copy_(torch.from_numpy(ho[:m])); d_dir[b:b + m].copy_(torch.from_numpy(hd[:m]))
        if b == 0 and keep_first:
            first = (ho[:min(m, keep_first)].copy(), hd[:min(m, keep_first)].copy())
    return d_org, d_dir, first


def record_views(torch, buf, m):
    """SoA views (prim i32, t, u, v f64) over one byte buffer of m * 28 bytes: t | u | v | prim"""
    t = buf[0:8 * m].view(torch.float64); u = buf[8 * m:16 * m].view(torch.float64)
    v = buf[16 * m:24 * m].view(torch.float64); p = buf[24 * m:28 * m].view(torch.int32)
    return (p, t, u, v)


def gather_ceiling(mb, mode, lds=40000, steps=200):
    """SURVEY 8d / VERDICT r04 item 4: what binds the incoherent walk is not the HBM datasheet figure but the memory system's rate
    for DEPENDENT random records -- tools/ubench/gather (one chain per lane, every link perturbed by the chain's own running
    sum), run here, on this box, at this leg's footprint and occupancy: mode 0 = 64-byte records (a 4-wide node), mode 5 =
    128-byte records (an 8-wide node).  -> {"records_per_s": G/s, ...} or {"error": ...}"""
    exe = os.path.join(ROOT, "tools", "ubench", "gather")
    cmd = [exe, str(int(mb)), str(steps), "4096", str(mode)]
    try:
        env = dict(os.environ, GATHER_LDS=str(lds))
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=env).stdout
        line = [l for l in out.splitlines() if l.startswith("array") and ("mode %d:" % mode) in l][-1]
        g = float(line.split("ms")[1].split("G chain-steps/s")[0])
        return {"records_per_s": round(g * 1e9, 0), "record_bytes": 64 if mode == 0 else 128, "footprint_MB": int(mb), "blocks_per_cu": int(160 * 1024 // lds),
                "cmd": "GATHER_LDS=%d tools/ubench/gather %d %d 4096 %d" % (lds, int(mb), steps, mode), "raw": line.strip(),
                "what": "dependent random gather, one chain per lane, 256 CUs x %d workgroups; the incoherent walk's binding ceiling (DESIGN 3.3)" % int(160 * 1024 // lds)}
    except Exception as e:                                  # noqa: BLE001 -- context, the leg stands without it
        return {"error": repr(e), "cmd": " ".join(cmd)}


def pmc_source(path, j):
    """where a `traffic` figure comes from: it is NOT measured inside this run (rocprofv3 counter passes re-run the whole
    command: tools/profile_round2.sh), it is the committed summary of the same command's last counter passes"""
    return {"file": path, "round": j.get("round"), "commit": j.get("commit"), "raw": j.get("source"),
            "FETCH_SIZE_KiB": j.get("FETCH_SIZE_KiB"), "WRITE_SIZE_KiB": j.get("WRITE_SIZE_KiB"),
            "formula": "2 x FETCH_SIZE x 1024 (gfx950: 128-B fabric requests tallied as 64 B) + WRITE_SIZE x 1024",
            "kernel_avg_ms_in_that_run": j.get("kernel_avg_ms_rocprof")}


def copy_rate(torch, dev):
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev); dst = torch.empty_like(big)
    dst.copy_(big); torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(big)
    e1.record(); torch.cuda.synchronize(dev)
    return 10 * 2 * big.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9       # read + write


def host_cores():
    """what this process may really use: os.cpu_count() is the box, the affinity mask and the cgroup CPU quota are the share"""
    n_os = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:                                            # noqa: BLE001
        aff = n_os
    quota = None
    try:                                                         # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:                                            # noqa: BLE001
        try:                                                     # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:                                        # noqa: BLE001
            quota = None
    eff = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return {"os_cpu_count": n_os, "sched_affinity": aff, "cgroup_cpu_quota": None if quota is None else round(quota, 2), "effective": eff}


def newest_pmc_summary(suffix, kernel_marker=None):
    """the counters of the newest profiles/r<NN>_<suffix> (a tools/pmc_cmd.sh summary): the LAST block whose header holds
    `kernel_marker` (None: the last block) -> (path relative to the repo, {counter: value}) or (None, None)"""
    import glob, re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_" + suffix)):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        return None, None
    blocks = []; cur = None
    for line in open(best[1]):
        if line.startswith("=="):
            cur = {"_header": line.strip()}; blocks.append(cur)
        elif cur is not None:
            m = re.match(r"\s+([A-Za-z0-9_]+)\s+([0-9.eE+-]+)\s*$", line)
            if m:
                cur[m.group(1)] = float(m.group(2))
    if kernel_marker is not None:
        blocks = [b for b in blocks if kernel_marker in b["_header"]] or blocks
    return (os.path.relpath(best[1], ROOT), blocks[-1]) if blocks else (None, None)
