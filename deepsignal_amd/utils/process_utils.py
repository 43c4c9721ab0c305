"""The slice of the reference's process_utils that is part of the call_mods I/O contract:
the 5-symbol base alphabet (reference deepsignal/utils/process_utils.py:21-22) and the
yes/no flag parser (:52-54)."""

base2code_dna = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}
code2base_dna = {v: k for k, v in base2code_dna.items()}


def str2bool(v) -> bool:
    return str(v).lower() in ("yes", "true", "t", "1")


def display_args(args) -> None:
    print("# ===============================================")
    print("## parameters: ")
    for key, val in vars(args).items():
        if key != "func":
            print("{}:\n\t{}".format(key, val))
    print("# ===============================================")
