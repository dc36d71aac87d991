"""File logger in ./log/ as the reference does (util/log.py:3-15)."""
import logging
import os


class Log:
    def __init__(self, module: str, filename: str):
        self.logger = logging.getLogger(module)
        self.logger.setLevel(logging.INFO)
        os.makedirs("./log/", exist_ok=True)
        handler = logging.FileHandler("./log/" + filename + ".log")
        handler.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
        self.logger.addHandler(handler)

    def add(self, text) -> None:
        self.logger.info(text)
