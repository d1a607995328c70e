#!/usr/bin/env python3
"""Build-time check (called by csrc/Makefile after the device TU is compiled): the register budget the frame times depend on.

Reads the code-object metadata of the -save-temps assembly (qr_device-hip-amdgcn-amd-amdhsa-gfx950.s) and fails the build when
  * the packet-walk instance qr_render_kernel<false,4,false> (every scene of the reference engine) spills a vector register, uses
    more than 128 VGPRs (4 waves per SIMD) or a private segment above the recursion frames' bytes (QR_MAX_SCRATCH, default 528);
  * the per-lane instance <false,3,true> exceeds 168 VGPRs (3 waves per SIMD) or spills more vector registers than QR_MAX_DIVK_SPILL;
  * the hand-written cull loop's fixed scalar registers s[88:99] (qr_walk.hpp cull_run) are missing from its clobber list.
usage: check_kernel_resources.py <file.s> [--print]
"""
import os, re, sys

SCR = int(os.environ.get("QR_MAX_SCRATCH", "528"))
LIMITS = {
    # mangled-name fragment: (max vgpr_count, max vgpr_spill_count, max private_segment_fixed_size)
    "16qr_render_kernelILb0ELi4ELb0EE": (128, 0, SCR),
    "22qr_render_multi_kernelILi4ELb0EE": (128, 0, SCR),
    "16qr_render_kernelILb0ELi3ELb1EE": (168, int(os.environ.get("QR_MAX_DIVK_SPILL", "24")), 640),
}
KEYS = ("name", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count")


def kernels(path):
    out, cur = [], None
    for line in open(path, errors="replace"):
        m = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "group_segment_fixed_size":
            cur = {}; out.append(cur)
        if cur is not None and k in KEYS:
            cur[k] = v if k == "name" else int(v)
    return [k for k in out if "name" in k and "vgpr_count" in k]


def main():
    path = sys.argv[1]
    ks = kernels(path)
    bad = []
    for k in ks:
        if "--print" in sys.argv:
            print({a: k.get(a) for a in KEYS})
        for frag, (vg, sp, scr) in LIMITS.items():
            if frag in k["name"]:
                if k["vgpr_count"] > vg: bad.append(f"{k['name']}: {k['vgpr_count']} VGPRs > {vg} (a wave per SIMD lost)")
                if k["vgpr_spill_count"] > sp: bad.append(f"{k['name']}: {k['vgpr_spill_count']} vector registers spilled > {sp}")
                if k["private_segment_fixed_size"] > scr: bad.append(f"{k['name']}: private segment {k['private_segment_fixed_size']} B > {scr}")
    missing = [frag for frag in LIMITS if not any(frag in k["name"] for k in ks)]
    if missing:
        bad.append(f"kernel instances missing from {path}: {missing}")
    src = os.path.join(os.path.dirname(os.path.abspath(path)), "qr_walk.hpp")
    if os.path.exists(src):
        text = open(src).read()
        if "cull_run" in text and "s88" in text:
            for r in range(88, 100):
                if not re.search(r'"s%d"' % r, text):
                    bad.append(f"qr_walk.hpp cull_run: s{r} is used by name but missing from the clobber list")
    if bad:
        print("KERNEL RESOURCE CHECK FAILED:\n  " + "\n  ".join(bad), file=sys.stderr)
        return 1
    print(f"kernel resource check ok ({len(ks)} kernels)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
