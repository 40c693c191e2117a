"""Early stopping on a sliding mean of the validation loss (reference ``h2gcn/modules/controller.py:4-30``)."""
from collections import deque


class SlidingMeanEarlyStopping:
    """``stopper(value)`` is True once the window is full and ``value`` exceeds the window's mean; otherwise the
    value enters the window (oldest leaves).  ``length == 0`` disables it."""

    def __init__(self, length: int):
        self.window = deque(maxlen=int(length))

    @property
    def length(self) -> int:
        return self.window.maxlen

    def reset(self) -> None:
        self.window.clear()

    def __call__(self, value) -> bool:
        value = float(value)
        if self.length <= 0:
            return False
        if len(self.window) == self.length and value > sum(self.window) / self.length:
            return True
        self.window.append(value)
        return False
