"""GPU: tensor parallelism of the harness on real kernels.

(a) TP=2 == TP=1: two fresh child processes, each one TP rank of the tiny model running the HIP kernels
    on cuda:0 (gloo carries the all-reduce / all-gather between them — two ranks cannot share one GPU
    under RCCL), produce the greedy tokens of the unsharded model: harness.py's sharding, collective
    placement and the column-parallel fusions (qkv_rope_cache, SILU epilogue) are what runs.
(b) RCCL inside the captured decode graph: a 1-rank NCCL (= RCCL) group with the collectives forced on;
    the decode step is captured into a HIP graph with the all-reduce / all-gather recorded in it and
    replayed; tokens equal the eager, collective-free run.  (More ranks need more GPUs: the 8-GPU curve
    is the driver's to measure.)
"""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "helpers", "tp_child.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, backend, graph, force=False, quant="awq"):
    port = _free_port()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tokens.json")
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), TP_BACKEND=backend, TP_GRAPH="1" if graph else "0", TP_OUT=out,
                       TP_FORCE_COLLECTIVES="1" if force else "0", TP_QUANT=quant)
            procs.append(subprocess.Popen([sys.executable, CHILD], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True))
        logs = [p.communicate(timeout=600)[0] for p in procs]
        for p, log in zip(procs, logs):
            assert p.returncode == 0, log[-3000:]
        with open(out) as f:
            return json.load(f)


def test_tp2_tokens_equal_tp1_on_hip_kernels():
    one = _run(1, "none", graph=False)
    two = _run(2, "gloo", graph=False)
    assert two["tokens"] == one["tokens"], (two["tokens"], one["tokens"])


def test_rccl_collectives_inside_the_captured_decode_graph():
    eager = _run(1, "none", graph=False)
    graphed = _run(1, "nccl", graph=True, force=True)
    assert graphed["graph"], f"decode graph was not captured: {graphed['graph_error']}"
    assert graphed["tokens"] == eager["tokens"]


def test_tp2_gptq_tokens_equal_tp1():
    one = _run(1, "none", graph=False, quant="gptq")
    two = _run(2, "gloo", graph=False, quant="gptq")
    assert two["tokens"] == one["tokens"]
