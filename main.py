#!/usr/bin/env python3
"""Entry point mirroring the reference CLI for the generation path: ``python main.py eval <config.yaml>``
(reference main.py:54-66).  train / export / download_data belong to subsystems outside the MI355X hot path."""
import sys


def main():
    if len(sys.argv) < 3 or sys.argv[1] != "eval":
        raise SystemExit("usage: python main.py eval <config.yaml>   (only the eval/generation path is provided)")
    from llamole_amd.eval import run_eval
    run_eval(sys.argv[2])


if __name__ == "__main__":
    main()
