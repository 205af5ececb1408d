"""Test infrastructure (build container only): dumps the reference's run.py CONFIG SURFACE as data.

For each BASELINE config (configs/config_{dvm_STiL,dvm_STiL_SAINT,cardiac_STiL}.yaml) the flat key/value namespace hydra
composes -- `defaults: [_self_, models: resnet50, dataset: <name>]` (configs/config_dvm_STiL.yaml:1-4; later entries win,
both group files are `# @package _global_`), with the dataset override of the README command (README.md:57) -- goes to
tests/golden/hparams_<config>.json, together with the set of keys the reference's STiL modules READ from it
(`self.hparams.<key>` / `args.<key>` in models/Disentangle/STiLModel*.py and utils/STiLModel*_backbone.py, run.py:29-98,
trainers/evaluate.py:142-166).  Only yaml values and attribute NAMES are stored -- no reference source text.

    python oracle/make_golden_hparams.py        # needs /root/reference; rewrites tests/golden/hparams_*.json
"""
import json
import os
import re
import sys

import yaml

REF = os.environ.get("STIL_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {  # config file -> (dataset yaml of the README-style command, the modules whose attribute reads are listed)
    "config_dvm_STiL": ("dvm_all_server_reordered_SemiPseudo_0.01",
                        ["models/Disentangle/STiLModel.py", "models/Disentangle/utils/STiLModel_backbone.py"]),
    "config_dvm_STiL_SAINT": ("dvm_all_server_reordered_SemiPseudo_0.01",
                              ["models/Disentangle/STiLModel_SAINT.py", "models/Disentangle/utils/STiLModel_SAINT_backbone.py"]),
    "config_cardiac_STiL": ("cardiac_CAD_SemiPseudo_0.01",
                            ["models/Disentangle/STiLModel.py", "models/Disentangle/utils/STiLModel_backbone.py"]),
}


def load(path):
    with open(path) as f:
        d = yaml.safe_load(f) or {}
    defaults = d.pop("defaults", [])
    return d, defaults


def compose(config, dataset):
    root, defaults = load(os.path.join(REF, "configs", config + ".yaml"))
    merged = {}
    for entry in defaults:          # hydra: in list order, later wins
        if entry == "_self_":
            merged.update(root)
        else:
            (group, name), = entry.items()
            if group == "dataset":
                name = dataset      # `dataset=<name>` on the command line (the yaml's own default file does not exist)
            d, _ = load(os.path.join(REF, "configs", group, name + ".yaml"))
            merged.update(d)
    return merged


def keys_read(files):
    keys = set()
    for f in files:
        keys |= set(re.findall(r"\b(?:hparams|args)\.([A-Za-z_][A-Za-z_0-9]*)", open(os.path.join(REF, f)).read()))
    return sorted(keys)


def main():
    for config, (dataset, files) in CASES.items():
        hp = compose(config, dataset)
        out = dict(config=config, dataset=dataset, hparams=hp, keys_read_by_reference=keys_read(files))
        path = os.path.join(OUT, f"hparams_{config}.json")
        json.dump(out, open(path, "w"), indent=1, sort_keys=True, default=str)
        print(path, len(hp), "keys;", len(out["keys_read_by_reference"]), "read on the path; algorithm_name =", hp.get("algorithm_name"))


if __name__ == "__main__":
    sys.exit(main())
