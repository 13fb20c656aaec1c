"""bench.py's implementation, split by leg (bench.py at the repo root is the entry point the driver runs)."""
from .common import parse, spawn_ranks, build_world, render_frame, pmc_traffic  # noqa: F401
from .cpu_leg import cpu_baseline  # noqa: F401
from .train_legs import train_leg, train_leg_sharded  # noqa: F401
from .frame import main  # noqa: F401
