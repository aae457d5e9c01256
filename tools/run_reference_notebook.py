#!/usr/bin/env python3
"""Run the code cells of one of the reference's notebooks, headless, against this repository's
`mppi_numba` alias package (the HIP engine + the reference's own host-side helpers).

    PYTHONPATH=/root/repo:/root/reference python tools/run_reference_notebook.py \
        /root/reference/test.ipynb [--cells 1,2,3,4] [--max-steps 20]

IPython magics are dropped, matplotlib draws to the Agg backend, `plt.show()` closes the
figures.  --max-steps rewrites `max_steps = <n>` assignments so that a closed loop can be cut
short.  Nothing of the notebook is stored here: it is read from the path given.
"""
import argparse
import json
import os
import re
import sys


def code_cells(path):
    with open(path) as fh:
        nb = json.load(fh)
    for index, cell in enumerate(nb["cells"]):
        if cell["cell_type"] == "code":
            yield index, "".join(cell["source"])


def sanitize(source, max_steps=None):
    lines = []
    for line in source.splitlines():
        if line.lstrip().startswith(("%", "!")):
            continue
        if max_steps is not None:
            line = re.sub(r"^(\s*max_steps\s*=\s*)\d+", lambda m: m.group(1) + str(max_steps), line)
        lines.append(line)
    return "\n".join(lines) + "\n"


def run(path, cells=None, max_steps=None, namespace=None):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.show = lambda *a, **k: plt.close("all")
    ns = namespace if namespace is not None else {"__name__": "__notebook__"}
    here = os.getcwd()
    os.chdir(os.path.dirname(os.path.abspath(path)))  # the notebooks use paths relative to themselves
    try:
        for index, source in code_cells(path):
            if cells is not None and index not in cells:
                continue
            if not source.strip():
                continue
            exec(compile(sanitize(source, max_steps), "%s[cell %d]" % (os.path.basename(path), index), "exec"), ns)
    finally:
        os.chdir(here)
    return ns


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("notebook")
    ap.add_argument("--cells", default=None, help="comma-separated cell indices (default: all code cells)")
    ap.add_argument("--max-steps", type=int, default=None)
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    cells = None if args.cells is None else {int(c) for c in args.cells.split(",")}
    ns = run(args.notebook, cells, args.max_steps)
    print("ran %s; names defined: %d" % (args.notebook, len(ns)))


if __name__ == "__main__":
    main()
