"""Tools keep the old convenience of steering an experiment from the shell — FT_REFILL_MIN, FT_MAX_BLOCKS_PER_CU, FT_HOST_CHUNKS,
FT_HOST_NO_PIN — but the LIBRARY no longer reads the environment: the tool reads the variable and sets the per-context option."""
import os

_ENV = {"FT_REFILL_MIN": ("refill_min", int), "FT_MAX_BLOCKS_PER_CU": ("max_blocks_per_cu", int), "FT_HOST_CHUNKS": ("host_chunks", int),
        "FT_HOST_NO_PIN": ("host_pin", lambda v: 0 if v not in ("", "0") else 1), "FT_TAIL_K": ("tail_k", int), "FT_MATH": ("math", int), "FT_GUIDED": ("guided", int), "FT_CULL": ("cull", int), "FT_ESCAPE": ("escape", int), "FT_LAZY_UNION": ("lazy_union", int), "FT_CARVED": ("carved", int), "FT_CHUNK": ("chunk", int), "FT_REUSE": ("reuse", int)}


def apply_env_options(dev):
    """-> {option: value} of what was applied"""
    done = {}
    for var, (name, conv) in _ENV.items():
        if var in os.environ and name in dev.OPTIONS:
            dev.set_option(name, conv(os.environ[var]))
            done[name] = dev.get_option(name)
    return done
