"""Same contract as the reference's rag/logging.py:1-9: a module-level `logger` at LOG_LEVEL."""
import logging
import os

logging.basicConfig(level=os.getenv("LOG_LEVEL", "INFO"),
                    format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
logger = logging.getLogger("rag")
