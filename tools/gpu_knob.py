import os, sys, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qr_loader import load_package
qr = load_package()
blob = gzip.decompress(open(os.path.join(ROOT,"tests","golden",sys.argv[1]+".qrs.gz"),"rb").read())
scn = qr.Scene(blob); f = scn.new_frame()
for _ in range(100): scn.render(f)
torch.cuda.synchronize()
avg, mn = scn.render_timed(f, 50)
print(sys.argv[1], "QR_DBG", os.environ.get("QR_DBG"), f"avg {avg*1e3:.1f} us min {mn*1e3:.1f} us")
