"""Host-side timeline marks (off unless ``start()`` was called): ``mark(name)`` appends (name, perf_counter) to the list that
``tools/e2e_timeline_probe.py`` folds into per-phase host times.  No GPU synchronisation -- the marks show where the HOST is."""
import time

events = None


def start():
    global events
    events = []


def stop():
    global events
    ev, events = events, None
    return ev


def mark(name: str):
    if events is not None:
        events.append((name, time.perf_counter()))
