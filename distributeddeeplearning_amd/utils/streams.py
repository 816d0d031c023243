"""HIP stream helpers."""
import os

import torch


def comm_stream(device=None):
    """A stream for collectives that should run BESIDE the compute stream's kernels.

    What round 3 measured (profiles/r3_dcgan28_forcedp_graph_timeline.md):
      * inside a captured step (the default) the graph executor gives parallel branches their own hardware queues: the
        bucket all-reduce runs beside the backward kernels whatever this stream's priority is;
      * in EAGER steps HIP multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES = 4) and two
        streams that share a queue execute in order: a default-priority communication stream landed on the compute
        stream's queue and overlapped nothing.  A high-priority stream (DDL_COMM_STREAM_PRIORITY=-1) gets a queue of
        its own and does overlap -- but it is NOT the default: with it the start-up tuner's RCCL candidates hung in a
        device synchronisation on the one-GPU box, and with several ranks sharing one GPU the hardware scheduler keeps
        one process's spinning all-reduce mapped ahead of the compute queue of the process it waits for."""
    prio = int(os.environ.get("DDL_COMM_STREAM_PRIORITY", "0"))
    return torch.cuda.Stream(device=device, priority=prio)
