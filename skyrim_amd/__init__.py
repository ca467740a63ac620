"""skyrim_amd: MI355X-native rollout engine behind the secondlaw-ai/skyrim API (Pangu hot path)."""
__version__ = "0.1.0"
