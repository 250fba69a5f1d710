from .drop_block import DropBlock2D

__all__ = ["DropBlock2D"]
