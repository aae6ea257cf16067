#!/usr/bin/env python3
"""oracle/gen_golden.py — regenerates tests/golden/ from the REFERENCE'S OWN code.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (this container): builds
oracle/_ref/ (make -C oracle ref: the reference's step13 / step1 translation units compiled
from where they lie, strict flags `g++ -O2 -mavx2 -mfma -ffp-contract=off`) and runs the
harness commands below.  The fixtures are DATA (inputs + outputs of the reference); no
reference source text goes into tests/golden/.

    python oracle/gen_golden.py            # rewrite every fixture
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
S13 = os.path.join(HERE, "_ref", "ref_step13_strict")
S1 = os.path.join(HERE, "_ref", "ref_step1_strict")

SEED = 10001


def run(*cmd):
    print("+", " ".join(str(c) for c in cmd), flush=True)
    out = subprocess.check_output([str(c) for c in cmd], cwd=GOLD).decode()
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
    manifest = {
        "generator": "oracle/gen_golden.py",
        "reference": "montib/rays1bench @ /root/reference, src/step13 + src/step1, compiled unmodified",
        "compiler": subprocess.check_output(["g++", "--version"]).decode().splitlines()[0],
        "flags_step13": "-O2 -mavx2 -mfma -ffp-contract=off -fno-rtti -fno-exceptions -std=c++17 -pthread -DNDEBUG",
        "flags_step1": "-O2 -ffp-contract=off -fno-rtti -fno-exceptions -std=c++17 -DNDEBUG",
        "seed": SEED,
        "files": {},
    }

    # (1) RNG known-answer vectors
    run(S13, "kat", "kat.bin")

    # (2) scene SoA + flat material table + camera, at the BASELINE aspect (1200x800),
    #     the reference's native aspect (1280x720) and config 1's (200x100)
    for scene in ("small", "medium", "large"):
        for (w, h) in ((1200, 800), (1280, 720), (200, 100), (80, 60)):
            run(S13, "scene", scene, w, h, f"scene_{scene}_{w}x{h}.bin")
    run(S13, "scene", "large", 1920, 1080, "scene_large_1920x1080.bin")
    run(S13, "scene", "large", 320, 200, "scene_large_320x200.bin")
    run(S13, "scene", "medium", 77, 45, "scene_medium_77x45.bin")
    run(S13, "scene", "small", 70, 50, "scene_small_70x50.bin")

    # (3) seeded per-sample goldens: 10k random (x, y, s) per scene at 1200x800x10
    for i, scene in enumerate(("small", "medium", "large")):
        run(S13, "samples", scene, 1200, 800, 10, SEED, 10000, 777 + i, f"samples_{scene}_1200x800x10.bin")
    #     + a second seed / size for the large scene
    run(S13, "samples", "large", 320, 200, 64, 424242, 4000, 99, "samples_large_320x200x64.bin")

    # (4) whole frames under the seeding contract (small sizes: image + ray totals)
    for scene in ("small", "medium", "large"):
        out = run(S13, "frame", scene, 200, 100, 4, SEED, 8, f"frame_{scene}_200x100x4.bin")
        manifest["files"][f"frame_{scene}_200x100x4.bin"] = json.loads(out)
    #     ragged size (not a multiple of the 32x32 tile, odd spp)
    out = run(S13, "frame", "medium", 77, 45, 3, 5, 8, "frame_medium_77x45x3.bin")
    manifest["files"]["frame_medium_77x45x3.bin"] = json.loads(out)

    # (5) the reference's deterministic single-thread path (sequential streams)
    for scene in ("small", "medium", "large"):
        out = run(S13, "seq", scene, 80, 60, 4, f"seq_{scene}_80x60x4.bin")
        manifest["files"][f"seq_{scene}_80x60x4.bin"] = json.loads(out)
    out = run(S13, "seq", "small", 70, 50, 3, "seq_small_70x50x3.bin")
    manifest["files"]["seq_small_70x50x3.bin"] = json.loads(out)

    # (6) BASELINE config 1: step1, small, 200x100, 1 spp
    print(subprocess.check_output([S1, "selfcheck"]).decode().strip().splitlines()[-1])
    out = run(S1, "frame1", "small", 200, 100, 1, "step1_small_200x100x1.bin")
    manifest["files"]["step1_small_200x100x1.bin"] = json.loads(out)
    out = run(S1, "frame1", "small", 64, 48, 3, "step1_small_64x48x3.bin")
    manifest["files"]["step1_small_64x48x3.bin"] = json.loads(out)

    # (7) full-size frames under the seeding contract: totals + image digests only
    #     (the 2.88 MB images are not committed; the GPU tests compare against the oracle
    #     live and against these totals)
    full = {}
    for scene in ("small", "medium", "large"):
        tmp = f"/tmp/r1_full_{scene}.bin"
        out = run(S13, "frame", scene, 1200, 800, 10, SEED, 8, tmp)
        rec = json.loads(out)
        sys.path.insert(0, HERE)
        import r1o  # noqa
        g = r1o.read_golden(tmp)
        rec["image_md5"] = hashlib.md5(g["image"].tobytes()).hexdigest()
        rec["image_mean"] = float(g["image"].astype("float64").mean())
        rec["rowrays"] = [int(v) for v in g["rowrays"]]
        full[scene] = rec
        os.remove(tmp)
    with open(os.path.join(GOLD, "full_1200x800x10.json"), "w") as f:
        json.dump(full, f, indent=0)

    # (8) the reference's native multi-threaded path at the BASELINE size, for the
    #     statistical comparison (not reproducible run to run; 3 runs recorded)
    stats = {}
    for scene in ("small", "medium", "large"):
        out = run(S13, "bench", scene, 1200, 800, 10, 8, 3)
        stats[scene] = [json.loads(l)["rays"] for l in out.strip().splitlines()]
    manifest["native_mt_rays_1200x800x10"] = stats

    for fn in sorted(os.listdir(GOLD)):
        if fn.endswith(".bin"):
            with open(os.path.join(GOLD, fn), "rb") as f:
                manifest["files"].setdefault(fn, {})["md5"] = hashlib.md5(f.read()).hexdigest()
    with open(os.path.join(GOLD, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("fixtures written to", GOLD)


if __name__ == "__main__":
    main()
