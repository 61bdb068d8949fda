"""MI355X-native RAD-NeRF head/torso frame renderer for GeneFace (see README.md)."""
import os as _os
import sys as _sys


def _runtime_is_up():
    """True / False when /proc tells whether this process already holds the compute driver's device node (the HIP runtime has started),
    None when it cannot be read."""
    try:
        fds = _os.listdir("/proc/self/fd")
    except OSError:
        return None
    for fd in fds:
        try:   # the listing contains its own, already closed, directory handle and any descriptor closed since: skip what no longer resolves
            if _os.readlink(f"/proc/self/fd/{fd}") == "/dev/kfd":
                return True
        except OSError:
            continue
    return False


def _more_hardware_queues():
    """The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The frame pipeline keeps several
    frames in flight on side streams next to the copy and default streams; with the default, a fourth frame in flight shares a hardware queue
    with something else and costs 5 % instead of gaining 2.5 % (NOTES.md section 5).  The variable is read when the runtime starts, so it is
    only set here if nobody has set it and the process has not opened the GPU yet (importing torch alone does not start the runtime).
    An embedding application that does not want its environment touched sets GENEFACE_AMD_KEEP_ENV=1 (or GPU_MAX_HW_QUEUES itself): the
    pipeline then keeps three frames in flight on the runtime's default queues."""
    if "GPU_MAX_HW_QUEUES" in _os.environ or _os.environ.get("GENEFACE_AMD_KEEP_ENV", "") not in ("", "0"):
        return
    up = _runtime_is_up()
    if up or (up is None and "torch" in _sys.modules):   # started, or cannot tell: leave the runtime's default
        return
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"


_more_hardware_queues()
