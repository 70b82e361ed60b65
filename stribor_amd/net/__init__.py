from .mlp import MLP
