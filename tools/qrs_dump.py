#!/usr/bin/env python3
"""Print a human-readable summary of a scene snapshot (include/qr_scene.h layout)."""
import struct, sys, gzip
import numpy as np

HDR = struct.Struct("<4I6I7I5I10I")

def load(path):
    raw = open(path, "rb").read()
    if path.endswith(".gz"):
        raw = gzip.decompress(raw)
    return raw

def main():
    raw = load(sys.argv[1])
    f = HDR.unpack_from(raw, 0)
    (magic, ver, total, hb, n_srf, n_mat, n_lgt, n_elm, n_tiles, n_tex,
     o_frame, o_srf, o_mat, o_lgt, o_elm, o_tiles, o_tex, s_frame, s_srf, s_mat, s_lgt, s_elm) = f[:22]
    print(f"magic {magic:08x} ver {ver} bytes {total} srf {n_srf} mat {n_mat} lgt {n_lgt} elm {n_elm} tiles {n_tiles} texels {n_tex}")
    fr = np.frombuffer(raw, dtype=np.uint8, count=s_frame, offset=o_frame)
    ff = fr.view(np.float32); fi = fr.view(np.int32)
    print("t_max", ff[0], "dir", ff[1:4], "hor", ff[4:7], "ver", ff[7:10])
    print("hor_a", ff[10:14], "ver_a", ff[14:18], "clamp", ff[18], "cmask", fi[19], "l_amb", ff[20], "amb", ff[21:24])
    print("t_min", ff[24], "org", ff[25:28], "flags", hex(fi[28]), "depth", fi[29], "fsaa", fi[30])
    print("frm", fi[31:34], "tile", fi[34:38], "clist", fi[38], "index/thnum", fi[39:41])
    srf = np.frombuffer(raw, dtype=np.int32, count=n_srf * 64, offset=o_srf).reshape(n_srf, 64)
    sf = srf.view(np.float32)
    for i in range(n_srf):
        s = srf[i]
        print(f"srf {i:3d} tag {s[37]:2d} t {s[34:37]} pos {sf[i,0:3]} trm {s[15]} shift {s[19]} axes {s[23]:03x} conic {s[11]} "
              f"mm {s[7]:02x} sci {sf[i,24:28]} scj {sf[i,28:31]} clip {s[38]} trn {s[39]} mat {s[40:42]} props {s[42]:04x} {s[43]:04x} lst {s[44:48]}")
    mat = np.frombuffer(raw, dtype=np.int32, count=n_mat * 32, offset=o_mat).reshape(n_mat, 32)
    mf = mat.view(np.float32)
    for i in range(n_mat):
        m = mat[i]
        print(f"mat {i:3d} scal {mf[i,0:2]} offs {mf[i,2:4]} mask {m[4:7]} tex {m[7]} tmap {m[8:10]} dff/spc {mf[i,10:12]} pow {m[12]} "
              f"rfl/trn/rfr {mf[i,13:16]} rfr2/rcp/ext2 {mf[i,16:19]}")
    lgt = np.frombuffer(raw, dtype=np.float32, count=n_lgt * 16, offset=o_lgt).reshape(n_lgt, 16)
    for i in range(n_lgt):
        print(f"lgt {i} {lgt[i,:12]}")
    elm = np.frombuffer(raw, dtype=np.int32, count=n_elm * 4, offset=o_elm).reshape(n_elm, 4)
    tiles = np.frombuffer(raw, dtype=np.int32, count=n_tiles, offset=o_tiles)
    print("tiles non-empty", int((tiles >= 0).sum()), "of", n_tiles)
    if len(sys.argv) > 2:
        for i in range(min(n_elm, int(sys.argv[2]))):
            print("elm", i, elm[i])

if __name__ == "__main__":
    main()
