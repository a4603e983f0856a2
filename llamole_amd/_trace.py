"""Timeline marks (off unless ``start()`` was called): ``mark(name)`` appends (name, perf_counter) to the list that ``bench.py`` folds into
per-phase host times under ``LLAMOLE_E2E_TRACE=1``.  No GPU synchronisation -- the marks show where the HOST is.  ``start(device=True)``
(``LLAMOLE_E2E_TRACE=2``) also records a HIP event on the CURRENT stream at every mark: where the DEVICE is on that stream."""
import time

events = None
device_events = None


def start(device: bool = False):
    global events, device_events
    events = []
    device_events = [] if device else None


def stop():
    global events, device_events
    ev, events = events, None
    dev, device_events = device_events, None
    return (ev, dev) if dev is not None else ev


def mark(name: str):
    if events is not None:
        events.append((name, time.perf_counter()))
        if device_events is not None:
            import torch
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            device_events.append((name, e))
