"""Compile every csrc/*.hip to gfx950 assembly (no GPU needed) and list each kernel's register / LDS / scratch use from the
code-object metadata, plus where its scratch (VGPR spill) accesses sit relative to the loops: spills in a prologue or epilogue
are harmless, spills inside a loop are what to look for before spending GPU time on a kernel change.
    python tools/check_spills.py [file.hip ...]        exit status 1 if any kernel has scratch traffic inside a loop"""
import glob
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llm-groundedvideodiffusion_amd", "csrc")
files = [os.path.abspath(f) for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()


def scratch_in_loops(lines):
    """{mangled kernel: (scratch accesses inside a backward-branch range, outside)}"""
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)] + [(len(lines), None)]
    out = {}
    for (a, name), (b, _) in zip(starts, starts[1:]):
        seg = lines[a:b]
        labels = {m.group(1): i for i, l in enumerate(seg) for m in [re.match(r"^(\.LBB\w+):", l)] if m}
        loops = [(labels[m.group(1)], i) for i, l in enumerate(seg) for m in [re.search(r"s_cbranch\w*\s+(\.LBB\w+)", l)]
                 if m and labels.get(m.group(1), i) < i]
        sc = [i for i, l in enumerate(seg) if "scratch_" in l]
        inside = sum(any(lo <= i <= hi for lo, hi in loops) for i in sc)
        out[name] = (inside, len(sc) - inside)
    return out


bad = 0
for f in files:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-unused-command-line-argument", "-S",
                        "--cuda-device-only", f, "-o", tmp.name], check=True, cwd=CSRC)
        text = open(tmp.name).read()
    where = scratch_in_loops(text.split("\n"))
    print(f"== {os.path.basename(f)}")
    for block in text.split("  - .agpr_count:")[1:]:
        get = lambda key: (re.search(rf"\.{key}:\s*(\S+)", block) or [None, "?"])[1]
        name = get("name")
        inside, outside = where.get(name, (0, 0))
        bad += inside > 0
        note = f"  scratch ops: {inside} in loops, {outside} outside" + ("  <-- SPILL IN A LOOP" if inside else "") if inside + outside else ""
        print(f"  vgpr {get('vgpr_count'):>4} agpr {block.split()[0]:>4} sgpr {get('sgpr_count'):>4} lds {get('group_segment_fixed_size'):>7} "
              f"scratch {get('private_segment_fixed_size'):>4} B  vgpr-spill {get('vgpr_spill_count'):>3} sgpr-spill {get('sgpr_spill_count'):>3}  "
              f"{demangle(name)[:96]}{note}")
sys.exit(1 if bad else 0)
