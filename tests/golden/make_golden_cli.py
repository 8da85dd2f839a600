"""Golden table of the reference's command lines: every flag of Classification/arg_parser.py and
DDPM/train.py with its default, obtained by running the reference's own parsers on an empty / minimal argv.

    python tests/golden/make_golden_cli.py
"""
from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    out = {}
    ap = MG._load("ref_arg_parser", MG.REF + "/Classification/arg_parser.py")
    argv, sys.argv = sys.argv, ["prog"]
    try:
        ns = ap.parse_args()
    finally:
        sys.argv = argv
    out["classification_defaults"] = {k: (v if isinstance(v, (int, float, str, bool, type(None))) else repr(v))
                                      for k, v in sorted(vars(ns).items())}
    # SD scripts: their parsers live under `if __name__ == "__main__"`, so the add_argument calls are read from the AST
    import ast

    def flags_of(path):
        tree = ast.parse(open(path).read())
        table = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args:
                name = ast.literal_eval(node.args[0])
                kw = {}
                for k in node.keywords:
                    if k.arg in ("default", "required"):
                        kw[k.arg] = ast.literal_eval(k.value)
                    elif k.arg == "type":
                        kw["type"] = getattr(k.value, "id", None)
                table[name.lstrip("-")] = kw
        return table

    out["sd"] = {nm: flags_of(MG.REF + "/SD/train-scripts/" + nm + ".py")
                 for nm in ("generate_mask", "random_label", "nsfw_removal", "proximal_gradient")}
    with open(os.path.join(HERE, "cli.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("cli.json written:", len(out["classification_defaults"]), "classification flags;",
          {k: len(v) for k, v in out["sd"].items()}, "SD flags")


if __name__ == "__main__":
    main()
