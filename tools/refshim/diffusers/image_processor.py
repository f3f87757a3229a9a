from typing import Any

PipelineImageInput = Any
