"""Root cause of the r03 abort of tests/test_segmented_gpu.py, isolated (VERDICT r03 #1 ii).

ProcessGroupNCCL's watchdog thread polls the end event of every collective with hipEventQuery.  On this HIP runtime the
query of an event fails with hipErrorCapturedEvent ("operation not permitted on an event last recorded in a capturing
stream") when the stream the event was recorded on is CAPTURING at the time of the query - also when the record itself
happened before the capture began (CUDA only refuses events recorded DURING a capture).  ddim._SegmentedForward issued its
warm-up forward - and with it RCCL calls - on the stream it captures on right afterwards: whenever the watchdog's 100-ms
poll fell between the last warm-up collective and its clean-up, the query raised inside the watchdog thread and
ProcessGroupNCCL terminated the process from there (SIGABRT wherever the main thread happened to be: torch.cat,
destroy_process_group).  This probe reproduces the runtime behaviour without RCCL: an event recorded on a stream BEFORE a
thread-local capture of that stream begins is queried from a second thread during the capture.

usage: python tools/diag/event_query_probe.py"""
import threading

import torch

dev = torch.device("cuda:0")
x = torch.zeros(1 << 20, device=dev)


def query_from_thread(ev):
    out = {}

    def run():
        try:
            out["done"] = ev.query()
        except Exception as exc:  # noqa: BLE001
            out["error"] = str(exc).splitlines()[0]
    th = threading.Thread(target=run)
    th.start()
    th.join()
    return out


for name, same in (("event recorded on ANOTHER stream than the one that captures", False),
                   ("event recorded on the stream that LATER captures", True)):
    side, other = torch.cuda.Stream(dev), torch.cuda.Stream(dev)  # fresh streams per case (a failed capture poisons its stream)
    rec_stream = side if same else other
    ev = torch.cuda.Event()
    with torch.cuda.stream(rec_stream):
        x.add_(1.0)
        ev.record(rec_stream)
    torch.cuda.synchronize()
    before = query_from_thread(ev)
    g = torch.cuda.CUDAGraph()
    during, capture = {}, "capture completed"
    try:
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            x.add_(1.0)
            during = query_from_thread(ev)
    except Exception as exc:  # noqa: BLE001 - the query from the other thread can also invalidate the capture itself
        capture = "capture FAILED: " + str(exc).splitlines()[0]
    torch.cuda.synchronize()
    after = query_from_thread(ev)
    print(f"{name}:\n   query before the capture {before}\n   query from a second thread DURING a thread-local capture of `side` {during}"
          f"\n   {capture}\n   query after {after}", flush=True)

import os, sys  # noqa: E402,E401
sys.stdout.flush()
os._exit(0)  # (the CUDAGraph object of the failed capture cannot be destroyed cleanly: torch terminates in its destructor)
