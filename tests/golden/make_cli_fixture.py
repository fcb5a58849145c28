"""Generates tests/golden/train_cli_flags.json: the command-line surface of the reference's train.py (flag names and literal
defaults), extracted by parsing /root/reference/train.py as text with `ast` (the file cannot be imported here: timm,
torchvision and tensorboardX are absent).  Only the flag table travels.  Run from the repo root:
    python tests/golden/make_cli_fixture.py"""
import ast
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/train.py"


def main():
    tree = ast.parse(open(REF).read())
    flags = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            names = [a.value for a in node.args if isinstance(a, ast.Constant) and isinstance(a.value, str)]
            if not names or not names[0].startswith("--"):
                continue
            entry = {}
            for kw in node.keywords:
                if kw.arg in ("default", "nargs", "action", "choices"):
                    try:
                        entry[kw.arg] = ast.literal_eval(kw.value)
                    except ValueError:
                        entry[kw.arg] = "<expr>"
                elif kw.arg == "type":
                    entry["type"] = getattr(kw.value, "id", None) or getattr(kw.value, "attr", "<expr>")
            flags[names[0]] = entry
    with open(os.path.join(HERE, "train_cli_flags.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_cli_fixture.py", "source": "reference train.py argument parser", "flags": flags},
                  f, indent=1, sort_keys=True)
    print(len(flags), "flags")


if __name__ == "__main__":
    main()
