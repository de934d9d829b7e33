"""`python -m variantcalling_amd <tool> [args]`: the tools of this package under the names the reference registers
(/root/reference/ugvc/__main__.py:42-56).  Each tool's `run` keeps its reference signature: the two filtering
pipelines take argv with the tool name first (as simppl hands it over), evaluate_concordance takes the bare flags
(/root/reference/ugvc/pipelines/evaluate_concordance.py:71,112-113); sec_training / correct_systematic_errors / sec_validation /
assess_sec_concordance are the four SEC tools of /root/reference/ugvc/__main__.py:19,44,56 (flags BUILDER-DEFINED: the reference
leaves them undocumented)."""
import sys

TOOLS = ("filter_variants_pipeline", "train_models_pipeline", "training_prep_pipeline", "evaluate_concordance", "calibrate_bridging_snvs",
         "sec_training", "correct_systematic_errors", "sec_validation", "assess_sec_concordance")


def main(argv):
    if len(argv) < 2 or argv[1] in ("-h", "--help") or argv[1] not in TOOLS:
        print("usage: python -m variantcalling_amd {" + ",".join(TOOLS) + "} [tool arguments]", file=sys.stderr)
        return 0 if len(argv) > 1 and argv[1] in ("-h", "--help") else 2
    import importlib
    mod = importlib.import_module(f"variantcalling_amd.pipelines.{argv[1]}")
    return mod.run(argv[2:] if argv[1] == "evaluate_concordance" else argv[1:]) or 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
