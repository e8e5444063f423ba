"""ISA of a lane-path model plugin with the product's flags (hipcc cross-compiles without a GPU): per-kernel instruction counts,
register use and spills, and optionally the whole text -- to check that an edit of csrc/dto_kkt_kernels.hpp leaves the hot
kernels' code generation alone (what decides k_kkt_fwd_seq's speed is where its spills land).

    python tools/isa_of_plugin.py acrobot [out.s]      # summary on stdout
"""
import os
import re
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def kernel_stats(isa):
    out = {}
    cur = None
    for ln in isa.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = dict(instructions=0, scratch=0, readlane=0, hash=0)
            continue
        if cur and ln.startswith("\t") and not ln.startswith("\t.") and not ln.startswith("\t;"):
            out[cur]["instructions"] += 1
            # (labels are numbered per function index in the file: normalised, so that the checksum only moves with the code)
            out[cur]["hash"] = zlib.crc32(re.sub(r"\.LBB\d+_", ".LBB_", ln.strip()).encode(), out[cur]["hash"])
            if "scratch_" in ln:
                out[cur]["scratch"] += 1
            if "v_readlane" in ln or "v_writelane" in ln:
                out[cur]["readlane"] += 1
        m = re.match(r"^\s*\.(sgpr_count|vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count):\s*(\d+)", ln)
        if m and cur:
            out[cur][m.group(1)] = int(m.group(2))
        m2 = re.match(r"^\s*\.name:\s*(\S+)", ln)
        if m2:
            cur = m2.group(1) if m2.group(1) in out else cur
    return out


def main():
    import check_exec_merge as C
    from dto_amd import plugin as PL, problems as P
    model = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
    p = getattr(P, f"build_{model}")(T=5, evaluate_hessian=True)
    st = PL.Structure(p["dynamics"], p["objective"], p["constraints"], None, True)
    src = PL.generate_source(st, model)
    path = os.path.join(PL.PLUGIN_DIR, f"_isa_{model}.hip")
    with open(path, "w") as f:
        f.write(src)
    isa = C.compile_to_isa(path, PL.BASE_CXXFLAGS + PL._extra_flags())
    os.remove(path)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(isa)
    # metadata block: per-kernel register counts
    meta = {}
    for m in re.finditer(r"\.agpr_count:\s*(\d+).*?\.name:\s*(\S+).*?\.sgpr_spill_count:\s*(\d+).*?\.vgpr_count:\s*(\d+).*?\.vgpr_spill_count:\s*(\d+)", isa, flags=re.S):
        meta[m.group(2)] = dict(agpr=int(m.group(1)), sgpr_spill=int(m.group(3)), vgpr=int(m.group(4)), vgpr_spill=int(m.group(5)))
    ks = kernel_stats(isa)
    for k in sorted(ks):
        if ks[k]["instructions"] < 50:
            continue
        short = re.sub(r"^_ZN?3dto\d*|I\d+.*$", "", k)[:40]
        print(f"{short:40s} instr {ks[k]['instructions']:6d} scratch {ks[k]['scratch']:4d} lane-ops {ks[k]['readlane']:5d} hash {ks[k]['hash']:08x} {meta.get(k, '')}")


if __name__ == "__main__":
    main()
