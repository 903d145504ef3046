"""quilt_amd -- MI355X-native engine for QUILT's per-sample HMM + Gibbs hot path."""
__version__ = "0.1.0"
