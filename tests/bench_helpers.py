"""Shared builders for GPU tests (small ensembles of the exported replica architectures)."""
import torch


def small_members(dp, n=3, batch=64, device="cuda:0"):
    """The first `n` exported replica networks (model0.. of the reference) as DP ensemble members + index batches."""
    from distributeddeeplearning_amd.bench_classifier import replica_specs
    from distributeddeeplearning_amd.config import get_space
    from distributeddeeplearning_amd.models.classifier import RandomModel
    from distributeddeeplearning_amd.models.optim import OptimConfig
    from distributeddeeplearning_amd.models.search import CompiledSpec
    from distributeddeeplearning_amd.utils.data import Data
    dev = torch.device(device)
    data = Data(dev, channels=1, n_train=512, n_test=16, synthetic=True, seed=0)
    space = get_space("ddl_normal4", ["data_train_size=512", f"batch_size={batch}"])
    members = []
    for i, spec in enumerate(replica_specs()[1:1 + n]):          # model1.. : the small ones
        compiled = CompiledSpec(spec, OptimConfig.keras("adam", 1e-3), "categorical_crossentropy")
        m = RandomModel(data, data.get_random_sample(512, seed=3), space, compiled, dev, number=i, seed=i, dp=dp,
                        use_graph=False, verbose=False)
        members.append(m)
    g = torch.Generator().manual_seed(7)
    idxs = [torch.randint(0, 512, (batch,), generator=g).to(dev) for _ in range(4)]
    return members, idxs
