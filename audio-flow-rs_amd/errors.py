"""AudioError -- mirror of the reference's error enum (src-tauri/src/error.rs:96-111)."""
from __future__ import annotations


class AudioError(Exception):
    """error.rs:96-111.  ``kind`` is the variant name; ``str()`` renders the
    reference's ``#[error(...)]`` text."""

    _FORMATS = {
        "NoDevice": "No input device available",                    # error.rs:97
        "ConfigurationFailed": "Device configuration failed: {0}",  # error.rs:100
        "StreamCreationFailed": "Stream creation failed: {0}",      # error.rs:103
        "CaptureFailed": "Capture failed: {0}",                     # error.rs:106
        "ResamplingFailed": "Resampling failed: {0}",               # error.rs:109
    }

    def __init__(self, kind: str, detail: str = "", status: int = 0):
        self.kind = kind
        self.detail = detail
        self.status = status
        super().__init__(self._FORMATS.get(kind, "{0}").format(detail))

    @classmethod
    def resampling_failed(cls, detail: str, status: int = 0) -> "AudioError":
        """resampler.rs:49,:85: every backend failure surfaces as ResamplingFailed(msg)."""
        return cls("ResamplingFailed", detail, status)

    def __eq__(self, other):
        return isinstance(other, AudioError) and (self.kind, self.detail) == (other.kind, other.detail)

    def __hash__(self):
        return hash((self.kind, self.detail))
