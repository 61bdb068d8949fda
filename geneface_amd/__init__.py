"""MI355X-native RAD-NeRF head/torso frame renderer for GeneFace (see README.md)."""
import os as _os
import sys as _sys


def _more_hardware_queues():
    """The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The frame pipeline keeps several
    frames in flight on side streams next to the copy and default streams; with the default, a fourth frame in flight shares a hardware queue
    with something else and costs 5 % instead of gaining 2.5 %.  The variable is read when the runtime starts, so it is
    only set here if nobody has set it and the process has not opened the GPU yet."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return
    try:   # the runtime is up once the process holds the compute driver's device node (torch.cuda.is_available() is enough to start it)
        if any(_os.readlink(f"/proc/self/fd/{fd}") == "/dev/kfd" for fd in _os.listdir("/proc/self/fd")):
            return
    except OSError:
        if "torch" in _sys.modules:   # cannot tell: leave the runtime's default, the pipeline then keeps three frames in flight
            return
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"


_more_hardware_queues()
