"""`python -m variantcalling_amd <tool> [args]`: the tools of this package under the names the reference registers
(/root/reference/ugvc/__main__.py:42-56).  Each tool's `run` keeps its reference signature: the two filtering
pipelines take argv with the tool name first (as simppl hands it over), evaluate_concordance takes the bare flags
(/root/reference/ugvc/pipelines/evaluate_concordance.py:71,112-113); sec_training / correct_systematic_errors / sec_validation /
assess_sec_concordance are the four SEC tools of /root/reference/ugvc/__main__.py:19,44,56 (flags BUILDER-DEFINED: the reference
leaves them undocumented).  `selftest [device]` is this package's own: the device canary (`ugvc_selftest`: a copy round trip and the
library's prefix-sum kernel against the host) - is this GPU usable at all, before a callset is committed to it."""
import sys

TOOLS = ("filter_variants_pipeline", "train_models_pipeline", "training_prep_pipeline", "evaluate_concordance", "calibrate_bridging_snvs",
         "sec_training", "correct_systematic_errors", "sec_validation", "assess_sec_concordance")


def selftest(argv):
    from .engine import Engine
    dev = int(argv[0]) if argv else 0
    with Engine(dev) as eng:
        info = eng.device_info()
        for n in (1, 64, 65, 1024, 1 << 20):
            eng.selftest(n)
    print(f"device {dev}: {info['name'].strip()}, {info['n_cus']} CUs, {info['hbm_bytes'] / 2**30:.0f} GiB: copies and computes")
    return 0


def main(argv):
    if len(argv) >= 2 and argv[1] == "selftest":
        return selftest(argv[2:])
    if len(argv) < 2 or argv[1] in ("-h", "--help") or argv[1] not in TOOLS:
        print("usage: python -m variantcalling_amd {" + ",".join(TOOLS) + ",selftest} [tool arguments]", file=sys.stderr)
        return 0 if len(argv) > 1 and argv[1] in ("-h", "--help") else 2
    import importlib
    mod = importlib.import_module(f"variantcalling_amd.pipelines.{argv[1]}")
    return mod.run(argv[2:] if argv[1] == "evaluate_concordance" else argv[1:]) or 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
