"""Data-parallel agent (reference: agent/barGen_horovod.py, which never ran: undefined
``dataset`` / ``make_batch`` and swapped ``hvd.DistributedOptimizer`` arguments).

Its intent -- one process per GPU, DistributedSampler sharding (:49-50), averaged gradient
all-reduce for the five optimizers (:91-99), initial broadcast of every state_dict (:130-134),
rank-0-only logging / summaries / checkpoints -- is what EVERY agent of this build does when it is
started under ``python -m torch.distributed.run`` (agent/base.py, hipops/dist.py over RCCL).  The
schedule of the Horovod file is the barGen_with_gan one, so this module is that agent."""
from agent.barGen_with_gan import BarGen  # noqa: F401
