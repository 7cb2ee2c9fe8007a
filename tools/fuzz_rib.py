"""Mutation fuzz of the RIB reader (lh_rib.c) under ASan + UBSan, on the CPU: truncations, byte flips, token splices and copied spans of
the example / parser-test RIBs must end in 0 or -1 with a message -- no crash, no sanitizer report, no leak.  Builds /tmp/rib_asan
from lh_rib.c alone.   python tools/fuzz_rib.py [first seed] [count]  (100 inputs per seed)"""
import glob, os, sys, subprocess, random
ROOT = "/root/repo"
srcs = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "rib", "*.rib"))) + sorted(glob.glob(os.path.join(ROOT, "tests", "ribparse", "*.rib")))
toks = [b"Polygon", b"PointsPolygons", b"\"P\"", b"[", b"]", b"WorldBegin", b"WorldEnd", b"AttributeBegin", b"AttributeEnd", b"Sphere", b"Translate", b"1e400", b"-", b"\"N\"", b"Format", b"Projection", b"\"perspective\"", b"\"fov\"", b"ConcatTransform", b"#", b"\"", b"nan", b"99999999999", b"Display", b"PointsGeneralPolygons", b"\"Cs\"", b"\"st\"", b"Orientation", b"\"lh\"", b"Sides", b"TransformBegin", b"TransformEnd", b"Scale", b"Rotate", b"Identity", b"PixelSamples", b"LightSource", b"Surface"]
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1; count = int(sys.argv[2]) if len(sys.argv) > 2 else 20; bad = 0
main_c = "/tmp/rib_asan_main.c"
open(main_c, "w").write('#include <stdio.h>\n#include "%s/include/lucille_hip.h"\nint main(int argc, char **argv) { for (int i = 1; i < argc; i++) { lh_rib_scene_t *sc = NULL; if (lh_rib_load(argv[i], &sc) == 0 && sc) { lh_rib_info_t inf; lh_rib_info(sc, &inf); lh_rib_free(sc); } } return 0; }\n' % ROOT)
subprocess.check_call(["gcc", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + ROOT + "/include", main_c, ROOT + "/lucille_amd/csrc/lh_rib.c", "-o", "/tmp/rib_asan", "-lm"])
os.makedirs("/tmp/ribfz", exist_ok=True)
for seed in range(seed0, seed0 + count):
    rnd = random.Random(seed); paths = []
    for k in range(100):
        data = bytearray(open(rnd.choice(srcs), "rb").read())
        for _ in range(rnd.randint(1, 8)):
            op = rnd.randint(0, 4); pos = rnd.randrange(len(data) + 1)
            if op == 0: data = data[:pos]
            elif op == 1 and data: data[rnd.randrange(len(data))] = rnd.randrange(256)
            elif op == 2: data[pos:pos] = rnd.choice(toks) + b" "
            elif op == 3 and len(data) > 10: a = rnd.randrange(len(data) - 5); del data[a:a + rnd.randint(1, 40)]
            elif op == 4 and len(data) > 10: a = rnd.randrange(len(data) - 5); data[pos:pos] = data[a:a + rnd.randint(1, 200)]
            if not data: data = bytearray(b" ")
        p = "/tmp/ribfz/s%d_%d.rib" % (seed, k); open(p, "wb").write(bytes(data)); paths.append(p)
    r = subprocess.run(["/tmp/rib_asan"] + paths, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"))
    if r.returncode != 0 or "runtime error" in r.stderr or "ERROR" in r.stderr:
        bad += 1; print("FINDING seed", seed, "rc", r.returncode); print(r.stderr[-1500:]); break
    for p in paths: os.remove(p)
print("%d x 100 mutated RIBs under ASan + UBSan, findings: %d" % (count, bad))
