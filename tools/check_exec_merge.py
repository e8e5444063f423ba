"""Finds, in gfx950 ISA, the code-generation fault behind the wrong-result modes of rounds 3-4:

    s_and_b64 exec, exec, <cond>        ; inner `if`: the mask is narrowed WITHOUT being saved -- the restore of the inner
    s_cbranch_execz .L_merge            ;   region was dropped because the enclosing region ends right behind it
      ... inner body ...
  .L_merge:
    v_accvgpr_read_b32 v52, a34         ; <- a reload the register allocator put into the merge block: it runs with the
    s_or_b64 exec, exec, s[0:1]         ;   INNER mask, every lane outside keeps a stale v52; the outer restore comes after

(AMD clang 22.0.0 / ROCm 7.2.0: SILowerControlFlow removes the "redundant" end-of-control-flow restore before register
allocation; later live-range splitting inserts vector instructions into the merge block.)  The build works around it with
`-mllvm -amdgpu-remove-redundant-endcf=0` (plugin.py BASE_CXXFLAGS, build.py); this tool checks any source / flag set:

    python tools/check_exec_merge.py file.hip [-- extra hipcc flags]     # compiles to ISA (device only) and scans it
    python tools/check_exec_merge.py file.s                              # scans ISA text

Exit code 1 if an instance is found.  `scan(text)` is what tests/test_exec_merge_guard.py calls.
"""
import os
import re
import subprocess
import sys
import tempfile

VEC = re.compile(r"^\s+(v_|ds_|global_|buffer_|scratch_|flat_)")
WRITES_EXEC = re.compile(r"^\s+s_\w+\s+exec\b|^\s+s_\w+saveexec")
NARROW = re.compile(r"^\s+s_and_b64\s+exec,\s*exec,")
BRANCH_Z = re.compile(r"^\s+s_cbranch_execz\s+(\S+)")
FUNC = re.compile(r"^(_Z\w+):")


def scan(text):
    """[(function, line number of the narrowing, merge label, [vector instructions executed under the narrowed mask])]"""
    lines = text.split("\n")
    labels = {}
    func_of = []
    cur = None
    for i, ln in enumerate(lines):
        m = FUNC.match(ln)
        if m:
            cur = m.group(1)
        func_of.append(cur)
        m = re.match(r"^(\.L\w+):", ln)
        if m:
            labels[(cur, m.group(1))] = i
    out = []
    for i, ln in enumerate(lines):
        if not NARROW.match(ln):
            continue
        # the conditional branch that goes with it (next instruction, comments / blank lines skipped)
        j = i + 1
        while j < len(lines) and (not lines[j].strip() or lines[j].lstrip().startswith(";")):
            j += 1
        m = BRANCH_Z.match(lines[j]) if j < len(lines) else None
        if not m:
            continue
        tgt = labels.get((func_of[i], m.group(1)))
        if tgt is None:
            continue
        bad = []
        k = tgt + 1
        while k < len(lines) and not WRITES_EXEC.match(lines[k]) and not lines[k].startswith(".Lfunc_end"):
            if VEC.match(lines[k]) and not re.match(r"^\s+v_(readlane|writelane|readfirstlane)", lines[k]):   # (lane ops ignore exec)
                bad.append((k + 1, lines[k].strip()))
            if re.match(r"^\s+s_(branch|cbranch|endpgm|setpc|swappc)", lines[k]):
                break
            k += 1
        if bad:
            out.append((func_of[i], i + 1, m.group(1), bad))
    return out


def compile_to_isa(src, flags):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(here, "directtrajectoryoptimization.jl_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", csrc,
               "-Wno-unused-value", "--cuda-device-only", "-S", "-o", out] + list(flags) + [src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        with open(out) as f:
            return f.read()


if __name__ == "__main__":
    args = sys.argv[1:]
    flags = []
    if "--" in args:
        k = args.index("--")
        args, flags = args[:k], args[k + 1:]
    rc = 0
    for fn in args:
        text = open(fn).read() if fn.endswith(".s") else compile_to_isa(fn, flags)
        hits = scan(text)
        print(f"{fn}: {len(hits)} merge block(s) with vector instructions under a narrowed exec mask "
              f"({sum(1 for ln in text.split(chr(10)) if NARROW.match(ln))} unsaved narrowings in all)")
        for fnname, ln, lab, bad in hits:
            print(f"  {fnname[:90]} line {ln} -> {lab}: " + "; ".join(f"{b[0]}: {b[1]}" for b in bad[:4]))
            rc = 1
    sys.exit(rc)
