"""Mirror of the reference's `rvc-common` crate: enums and the error type.

Reference: /root/reference/rvc-common/src/enums.rs:4-146, rvc-common/src/errors.rs:2-20.
"""
from __future__ import annotations

import enum


class RvcModelVersion(enum.Enum):
    V1 = 1
    V2 = 2

    def text_encoder_in_channels(self) -> int:      # enums.rs:10-16
        return 256 if self is RvcModelVersion.V1 else 768

    def output_layers(self) -> int:                 # enums.rs:17-23
        return 9 if self is RvcModelVersion.V1 else 12

    @classmethod
    def from_value(cls, val) -> "RvcModelVersion":
        """enums.rs:42-74: 1/"v1" -> V1, 2/"v2" -> V2, anything else silently maps to V2."""
        if isinstance(val, cls):
            return val
        if val in (1, "v1"):
            return cls.V1
        return cls.V2

    def __int__(self) -> int:                        # enums.rs:32-40
        return self.value

    def __str__(self) -> str:                        # enums.rs:53-60, 76-83
        return "v1" if self is RvcModelVersion.V1 else "v2"

    @staticmethod
    def is_valid(val: int) -> bool:                  # enums.rs:85-92
        return val in (1, 2)


class PitchAlgorithm(enum.Enum):
    Rmvpe = 1

    @classmethod
    def from_value(cls, val) -> "PitchAlgorithm":
        """enums.rs:104-133: every value maps to Rmvpe."""
        return cls.Rmvpe

    def __int__(self) -> int:
        return 1

    def __str__(self) -> str:
        return "rmvpe"

    @staticmethod
    def is_valid(val: int) -> bool:
        return val == 1


class RvcInferError(Exception):
    """errors.rs:2-8.  `kind` is one of the variant names; Ort(..) is called Backend here."""

    KINDS = {1: "ModelNotLoaded", 2: "ContentvecNotLoaded", 3: "F0NotLoaded", 4: "Backend", 5: "NdarrayShapeError", 6: "Panic"}

    def __init__(self, code: int, message: str = ""):
        self.code = code
        self.kind = self.KINDS.get(code, "Unknown(%d)" % code)
        super().__init__("%s%s" % (self.kind, (": " + message) if message else ""))
