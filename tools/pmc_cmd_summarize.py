"""per kernel (ray-query kernels only): the counters of its largest dispatch in each pass of tools/pmc_cmd.sh"""
import csv, glob, re, sys, collections
d = sys.argv[1]
best = collections.OrderedDict()          # kernel -> counter -> value of the dispatch with the largest SQ_WAVES-independent proxy
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(dict)   # (kernel, dispatch) -> counter -> value
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("k_trace", "k_resolve", "k_quad", "k_pt_decide", "k_pt_scatter", "k_pt_resolve", "k_ao_setup")):
            continue
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    # largest dispatch per kernel = the one with the largest first counter
    bykernel = collections.defaultdict(list)
    for (k, disp), c in per.items():
        bykernel[k].append(c)
    for k, lst in bykernel.items():
        top = max(lst, key=lambda c: max(c.values()))
        best.setdefault(k, {}).update(top)
for k, c in best.items():
    nm = re.search(r"(k_\w+(?:<[^>]*>)?)", k)
    print("==", nm.group(1) if nm else k[:90])
    for name, v in c.items():
        print("   %-28s %.6g" % (name, v))
    g = c.get
    if g("SQ_WAVE_CYCLES"):
        print("   -> wait_any / wave_cycles          %.3f" % (g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")))
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        print("   -> VALU lane utilisation           %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)))
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_BUSY_CYCLES"):
        print("   -> VALU pipe busy (x4 / (4 SIMD x busy cycles))  %.3f" % (g("SQ_ACTIVE_INST_VALU") * 4 / (4.0 * g("SQ_BUSY_CYCLES"))))
    if g("TCC_REQ_sum"):
        print("   -> L2 hit rate                     %.3f   EA read requests x 128 B = %.2f GB" % (g("TCC_HIT_sum", 0) / (g("TCC_HIT_sum", 0) + g("TCC_MISS_sum", 1)), g("TCC_EA0_RDREQ_sum", 0) * 128 / 1e9))
