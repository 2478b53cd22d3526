"""Register / scratch / instruction-mix figures of the kernels of one translation unit, compiled with the library's flags (no GPU needed).
usage: python tools/isa_stats.py emcee_amd/csrc/emx_slab.hip [regex on the demangled name] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emcee_amd import _build  # noqa: E402


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    src = args[0]
    filt = re.compile(args[1] if len(args) > 1 else ".")
    flags = _build.FLAGS + _build.EXTRA_FLAGS.get(os.path.basename(src), []) + extra
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-3000:])
            sys.exit(1)
        text = open(out).read()
    meta = {}
    for blk in re.split(r"\n  - \.agpr_count:", text)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in
                      ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size")}
        meta[name]["agpr_count"] = int(blk.split()[0])
    names = list(meta)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    for name, d in zip(names, dem):
        d = re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "")).replace("emx::", "")
        if not filt.search(d):
            continue
        m = re.search(r"^%s:[^\n]*\n(.*?)s_endpgm" % re.escape(name), text, re.S | re.M)
        body = m.group(1) if m else ""
        cnt = lambda pat: len(re.findall(pat, body))
        s = meta[name]
        print("%-48s vgpr %3d agpr %3d sgpr %3d scratch %4d B  spilled v/s %d/%d | mfma %d, scratch ld/st %d/%d, global ld/st %d/%d, buffer ld/st %d/%d, ds rd/wr %d/%d, s_waitcnt %d"
              % (d, s["vgpr_count"], s["agpr_count"], s["sgpr_count"], s["private_segment_fixed_size"], s["vgpr_spill_count"], s["sgpr_spill_count"],
                 cnt(r"v_mfma"), cnt(r"scratch_load"), cnt(r"scratch_store"), cnt(r"global_load"), cnt(r"global_store"), cnt(r"buffer_load"),
                 cnt(r"buffer_store"), cnt(r"ds_read|ds_load"), cnt(r"ds_write|ds_store"), cnt(r"s_waitcnt")))


if __name__ == "__main__":
    main()
