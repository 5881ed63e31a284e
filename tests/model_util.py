"""Re-export of the product's synthetic harness for the tests."""
from mgnns_amd.harness import build_model, call_args, make_vocab, synthetic_adjacencies  # noqa: F401
