"""HIP-graph replay of the dense segments (distributed/hip_graph.py, DLRMTrain.capture_hip_graphs):
a graphed train loop must give the eager loop's losses and parameters."""
import numpy as np
import pytest
import torch

import _paths  # noqa: F401

pytestmark = pytest.mark.gpu


def _train(hip_graphs: bool, steps: int = 6, flat: bool = False):
    from torchrec_amd.datasets.random import RandomRecDataset
    from torchrec_amd.distributed.embeddingbag import EmbeddingBagCollectionSharder
    from torchrec_amd.distributed.model_parallel import DistributedModelParallel
    from torchrec_amd.distributed.train_pipeline import TrainPipelineSparseDist
    from torchrec_amd.distributed.types import ShardingEnv
    from torchrec_amd.models.dlrm import DLRMTrain
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig
    from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection
    from torchrec_amd.optim.keyed import CombinedOptimizer, KeyedOptimizerWrapper

    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    rows, D, B, lr = [1000, 3, 57, 20000, 11], 128, 2048, 0.05
    keys = [f"cat_{i}" for i in range(len(rows))]
    tables = [EmbeddingBagConfig(name=f"t_{k}", embedding_dim=D, num_embeddings=rows[i], feature_names=[k])
              for i, k in enumerate(keys)]
    ebc = EmbeddingBagCollection(tables, device=torch.device("meta"))
    tm = DLRMTrain(ebc, 13, [512, 256, D], [1024, 512, 1], dense_device=dev)
    model = DistributedModelParallel(tm, env=ShardingEnv.from_local(1, 0), device=dev,
                                     sharders=[EmbeddingBagCollectionSharder({"learning_rate": lr})])
    data = RandomRecDataset(keys, B, rows, manual_seed=5, num_generated_batches=4, num_batches=steps + 2, device=dev)
    if flat:  # gradients of the graphed segments through one flat buffer (what bench.py does for N > 1)
        tm.capture_hip_graphs(B, flat_grads=True)
    # as bench.py: the dense optimizer is built after the capture; in flat mode it is the one-kernel FlatSGD
    dense_opt = KeyedOptimizerWrapper(dict(model.named_parameters()), lambda p: tm.dense_optimizer(p, lr=lr))
    from torchrec_amd.optim.flat import FlatSGD
    assert isinstance(dense_opt._optimizer, FlatSGD) == flat
    opt = CombinedOptimizer([model.fused_optimizer, dense_opt])
    pipe = TrainPipelineSparseDist(model, opt, dev, hip_graphs=hip_graphs and not flat)
    model.train()
    it = iter(data)
    losses = [float(pipe.progress(it)[0].detach()) for _ in range(steps)]
    # graph mode under the pipeline (captured here in flat-gradient mode, or lazily by the pipeline) = the explicit
    # (no autograd engine) step, also without an exchange
    assert getattr(tm, "explicit_steps", 0) == (steps if (flat or hip_graphs) else 0)
    # one more step with ANOTHER batch size: graphed models must fall back to the eager segments (and, in
    # flat-gradient mode, still deliver the gradients through the flat buffer)
    small = RandomRecDataset(keys, B // 2, rows, manual_seed=9, num_generated_batches=1, num_batches=1, device=dev)
    pipe._requests.clear()  # drop the queued input_dist of the next (full-size) batch: this step is manual
    opt.zero_grad()
    loss, _ = model(next(iter(small)))
    loss.backward()
    tm.finish_dense_grads()
    opt.step()
    losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    assert (model.module._graphs is not None) == hip_graphs
    params = {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}
    shards = {n: w.detach().cpu().numpy().copy() for n, (w, _) in model.sharded_modules()[0].local_shards().items()}
    return losses, params, shards, model, pipe, data


@pytest.mark.parametrize("flat", [False, True])
def test_graphed_train_loop_matches_eager(flat):
    l0, p0, s0, *_ = _train(False)
    l1, p1, s1, model, pipe, data = _train(True, flat=flat)
    np.testing.assert_allclose(l1, l0, rtol=1e-5, atol=1e-6)
    for k in p0:
        np.testing.assert_allclose(p1[k], p0[k], rtol=1e-4, atol=1e-6, err_msg=k)
    for k in s0:
        np.testing.assert_allclose(s1[k], s0[k], rtol=1e-4, atol=1e-6, err_msg=k)
    # eval / no_grad steps fall back to the eager path and leave the graphs intact
    model.eval()
    with torch.no_grad():
        loss, _ = model(next(iter(data)))
    assert np.isfinite(float(loss))
