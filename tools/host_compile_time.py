#!/usr/bin/env python3
"""Host-only: per-phase time of validate + compile of a snapshot (QR_COMPILE_TIMING ticks of csrc/qr_compile.cpp), and a hash
of the compiled image (QR_DUMP_IMAGE) so that a change to the compiler can be checked to leave the image byte for byte.
usage: tools/host_compile_time.py [snapshot.qrs.gz | synth:N:W:H:DEPTH] [reps]"""
import gzip, hashlib, importlib, os, re, subprocess, sys, tempfile

def main():
    snap = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/c2b_demo01_1080p.qrs.gz"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if os.environ.get("_QR_CHILD"):
        q = importlib.import_module("quadray-engine_amd")
        if snap.startswith("synth:"):
            import bench
            blob = bytes(bench.load_blob(snap))
        else:
            blob = gzip.open(snap).read()
        for _ in range(reps):
            q.program_stats(blob)
        return
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, _QR_CHILD="1", QR_COMPILE_TIMING="1", QR_DUMP_IMAGE=os.path.join(d, "img"))
        r = subprocess.run([sys.executable, __file__, snap, str(reps)], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr)
        ph = {}
        for m in re.finditer(r"compile (\S+)\s+([0-9.]+) ms", r.stderr):
            ph.setdefault(m.group(1), []).append(float(m.group(2)))
        tot = 0.0
        for k, v in ph.items():
            v.sort(); med = v[len(v) // 2]; tot += med
            print(f"{k:10s} {med:.3f} ms")
        print(f"{'sum':10s} {tot:.3f} ms   image sha1 {hashlib.sha1(open(os.path.join(d, 'img'), 'rb').read()).hexdigest()[:16]}")

if __name__ == "__main__":
    main()
