"""Stand-in for the third-party `loguru` package (absent from this image): a logger object with the usual level methods."""
import logging as _pylog


class _Logger:
    _log = _pylog.getLogger("loguru-shim")

    def info(self, msg, *a, **k):
        self._log.info(msg)

    def debug(self, msg, *a, **k):
        self._log.debug(msg)

    def warning(self, msg, *a, **k):
        self._log.warning(msg)

    def error(self, msg, *a, **k):
        self._log.error(msg)

    def exception(self, msg, *a, **k):
        self._log.error(msg)


logger = _Logger()
