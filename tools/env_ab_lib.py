"""set_knob(): what tools/env_ab.py and tools/fwd_ab.py flip between their interleaved runs."""
import os


def set_knob(name, value):
    """VP3D_* -> environment variable (read at call time by the package); SW:key -> videopose3d_amd._switches.SW[key]
    ("0" / "1" become False / True where the switch is boolean)."""
    if name.startswith("SW:"):
        from videopose3d_amd import ops_s16
        from videopose3d_amd._switches import SW
        key = name[3:]
        SW[key] = (value != "0") if isinstance(SW[key], bool) else value
        ops_s16._plan_cache.clear()
        ops_s16._red_ok.clear()
    else:
        os.environ[name] = value
