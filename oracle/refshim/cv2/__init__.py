"""cv2 is absent here; the hot path never calls it."""
__version__ = "4.0.0"
