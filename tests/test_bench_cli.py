"""bench.py contract, graft entry, CLI smoke runs, small utilities."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config"}


def _run(args, **kw):
    return subprocess.run([sys.executable] + args, cwd=kw.pop("cwd", ROOT), capture_output=True, text=True,
                          timeout=600, **kw)


@pytest.mark.parametrize("cfg", ["dcgan28", "classifier"])
def test_bench_contract_cpu(cfg):
    r = _run(["bench.py", "--device", "cpu", "--steps", "2", "--warmup", "1", "--batch", "8", "--config", cfg])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    assert d["config"]["parallelism"] == "dp1" and d["data"] == "synthetic"


def test_bench_force_dp_one_rank_cpu():
    """--force-dp runs the data-parallel code path (collectives on a 1-rank gloo group) in one process."""
    env = dict(os.environ, MASTER_PORT="29641")
    r = _run(["bench.py", "--device", "cpu", "--steps", "2", "--warmup", "1", "--batch", "8", "--force-dp"], env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0


def test_bench_two_ranks_gloo():
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", "bench.py", "--gpus", "2",
                        "--device", "cpu", "--steps", "2", "--warmup", "1", "--batch", "8", "--config", "localsgd",
                        "--avg-every", "2"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 80


def test_bench_two_ranks_gloo_classifier_reports_comm_fields():
    """Sync-DP classifier (ReplicaGroupStep) on 2 gloo ranks; the record proves what the run really used."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29633", "bench.py", "--gpus", "2",
                        "--device", "cpu", "--steps", "2", "--warmup", "1", "--batch", "8", "--config", "classifier"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == [0, 1] and d["graph_mode_used"] == "off"
    assert d["comm_backend"] == "torch.distributed/gloo" and d["captured_collective_selftest"] is None


def test_graft_entry_build():
    r = _run(["-c", "import __graft_entry__ as g; g.build()"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "libddl_hip.so" in r.stdout


def test_cli_ddl_and_gan_smoke(tmp_path):
    r = _run([os.path.join(ROOT, "entrypoints", "DDL_Normal3.py"), "--device", "cpu", "--synthetic", "--train-size",
              "400", "--set", "data_train_size=300", "--set", "random_model_size=2", "--set", "transfer_model_size=1",
              "--set", "predict_model_size=2", "--set", "epochs=2", "--set", "conv2d_size=0,2", "--quiet", "--save"],
             cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("model accuracy : ") == 3 and "ensemble accuracy : " in r.stdout
    assert (tmp_path / "models" / "model0" / "weights.h5").exists()
    r = _run([os.path.join(ROOT, "entrypoints", "GAN_distributed.py"), "--device", "cpu", "--synthetic", "--set",
              "iterations=2", "--set", "train_images=8", "--set", "log_period=1", "--set", "sample_images=2"],
             cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "generator loss : " in r.stdout and "discriminator loss : " in r.stdout
    assert (tmp_path / "images" / "1.jpg").exists() and (tmp_path / "generator.png").exists()


def test_export_five_untrained_models_like_main2(tmp_path):
    r = _run([os.path.join(ROOT, "entrypoints", "DDL_Normal4.py"), "--device", "cpu", "--synthetic", "--train-size",
              "200", "--quiet"], cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    from distributeddeeplearning_amd.utils import ckpt
    names = set()
    for i in range(5):
        net = ckpt.load_model(tmp_path / "models" / f"model{i}", "cpu")
        names |= {l.name for l in net.spec.layers}
        assert net.spec.output_shape == (10,)
    assert "input_5" in names and "dense_5" in names            # Keras-style global name counters


def test_image_plot_meter_utils(tmp_path):
    from distributeddeeplearning_amd.models.dcgan import generator_spec
    from distributeddeeplearning_amd.utils import image, plot, trace
    a = image.array_to_uint8(np.array([[0.2, 0.4], [0.6, 1.0]]))
    assert a.min() == 0 and a.max() == 255                       # keras save_img scale=True: min-max rescale
    p = image.save_img(tmp_path / "x.jpg", np.random.rand(28, 28, 1))
    from PIL import Image
    assert Image.open(p).size == (28, 28)
    dot = plot.to_dot(generator_spec())
    assert "conv2d_transpose_1: Conv2DTranspose" in dot and "(None, 14, 14, 64)" in dot
    m = trace.StepMeter(100)
    for _ in range(3):
        with m:
            pass
    assert m.steps == 3 and (m.images_per_sec() > 0 or torch.cuda.is_available())
    with trace.range("x"):
        pass
    lg = trace.JsonlLogger(str(tmp_path / "l.jsonl"))
    lg.log(step=1, loss=0.5)
    lg.close()
    assert json.loads(open(tmp_path / "l.jsonl").read())["loss"] == 0.5
