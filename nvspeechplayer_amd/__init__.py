"""nvspeechplayer_amd -- MI355X-native Klatt synthesis engine behind the speechPlayer C-ABI."""
from . import _native  # noqa: F401
from .speechPlayer import BatchPlayer, Frame, LiveGroup, NodePlayer, SpeechPlayer, host_array, setGlobalOption  # noqa: F401

__all__ = ["Frame", "SpeechPlayer", "BatchPlayer", "NodePlayer", "LiveGroup", "host_array", "setGlobalOption"]
