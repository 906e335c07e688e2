"""Development repro: 2 ranks on one GPU (gloo), DLRM train loop with HIP-graph segments under DDP."""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def worker(rank, W, port):
    faulthandler.enable()
    import test_multirank_gpu as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    T._stage_a2a_through_host()
    from torchrec_amd.distributed.types import ShardingEnv
    import torchrec_amd.distributed.train_pipeline as tp
    keys, model, opt = T._e2e_model(ShardingEnv.from_process_group(dist.group.WORLD), dev, dp_max_rows=10)
    T._e2e_init_tables(model)
    # force graphs on under DDP
    orig = tp.TrainPipelineSparseDist.__init__

    def init(self, m, o, d, hip_graphs=False):
        orig(self, m, o, d, hip_graphs=False)
        self._hip_graphs = True
    tp.TrainPipelineSparseDist.__init__ = init
    if os.environ.get("SERIALIZE_CAPTURE") == "1":  # one rank captures at a time (they share the GPU here)
        from torchrec_amd.models.dlrm import DLRMTrain
        cap = DLRMTrain.capture_hip_graphs

        def serial(self, B):
            for r in range(W):
                if r == rank:
                    cap(self, B)
                torch.cuda.synchronize()
                dist.barrier()
        DLRMTrain.capture_hip_graphs = serial
    print(rank, "start", flush=True)
    out = T._e2e_run(model, opt, keys, T._e2e_batches(W), rank, W, dev, True)
    print(rank, "losses", out[0], flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29533), nprocs=2, join=True)
