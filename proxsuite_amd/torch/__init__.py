"""`proxsuite.torch` of the reference: the QPLayer forward on the MI355X batch solver."""
from .qplayer import QPFunction

__all__ = ["QPFunction"]
