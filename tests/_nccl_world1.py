"""Worker of tests/test_gpu_parity.py::test_gpu_rccl_communicator_beside_the_hip_library: RCCL (`nccl` backend) with world
size 1 beside libqrhip.so in one process; a HIP-rendered frame through the collectives and through sharding.py's paths."""
import importlib.util
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from qr_loader import load_package
    from conftest import load_blob, load_frame
    qr = load_package()
    spec = importlib.util.spec_from_file_location("qr_sharding", os.path.join(ROOT, "quadray-engine_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)

    torch.cuda.set_device(0)
    scn = qr.Scene(load_blob("demo01_160"))                   # HIP kernels of libqrhip.so first ...
    frame = scn.render(); torch.cuda.synchronize()
    want = torch.from_numpy((load_frame("demo01_160") & 0xFFFFFF).astype("int64")).to(torch.int32).cuda()
    assert bool((frame == want).all())
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))     # ... then an RCCL communicator in the same process
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    got = torch.empty_like(frame)
    dist.all_gather_into_tensor(got, frame)                   # a frame-sized all_gather (SURVEY 8e's collective)
    red = frame.clone(); dist.all_reduce(red, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    assert bool((got == want).all()) and bool((red == want).all())
    h, w = frame.shape
    ex = sharding.FrameExchange(h, w, 1, 0)
    fin = torch.zeros_like(frame); ex.exchange_many([([frame], fin)])
    gf = [torch.zeros_like(frame)]; ex.gather_many([([frame], gf)], root=0)
    sp = sharding.SplitFrame(h, w, 1, 0)
    buf = torch.zeros((sp.alloc_rows, w), dtype=torch.int32, device="cuda"); buf[:h] = frame
    sfin = torch.zeros_like(buf); sp.gather([(buf, sfin)], root=0)
    again = scn.render(); torch.cuda.synchronize()            # and the renderer still works beside the communicator
    assert bool((fin == want).all()) and bool((gf[0] == want).all()) and bool((sfin[:h] == want).all()) and bool((again == want).all())
    dist.barrier()
    dist.destroy_process_group()
    print("rccl ok: backend nccl, world 1,", torch.cuda.get_device_name(0))


if __name__ == "__main__":
    main()
