"""Two ranks on ONE GPU (all a gpurun box offers): can the library's RCCL communicator be initialised and reduce?  RCCL normally refuses
two ranks on the same device ("Duplicate GPU detected"); this prints what happens here.  torchrun-free: spawn + gloo for the id exchange."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hulc_amd import spec
    from hulc_amd.engine import StepEngine
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    eng = StepEngine(dims, 2, 4, dtype="bf16", device="cuda:0", dropout_p=0.0)
    eng.load_numpy(spec.init_all(dims, seed=0))
    box = [eng.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    try:
        eng.comm_init(box[0], rank, world)
        eng.flat_grads.fill_(float(rank + 1))
        eng.allreduce_grads("fp32")
        torch.cuda.synchronize()
        print(f"rank {rank}: all-reduce ok, value {eng.flat_grads[12345].item()} (expect 3.0)", flush=True)
    except Exception as e:
        print(f"rank {rank}: {type(e).__name__}: {e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29733), nprocs=2, join=True)
