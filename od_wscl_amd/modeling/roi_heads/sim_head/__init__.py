from .sim_net import Sim_Net
from .sim_loss import SupConLossV2

__all__ = ["Sim_Net", "SupConLossV2"]
