"""scopa_amd -- MI355X-native MiniScopa CFR traversal engine (drop-in solver backend for rug-marl-group2/scopa)."""
__version__ = "0.1.0"
