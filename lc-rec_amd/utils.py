"""Small host helpers with the behaviour of the reference's index/utils.py (directory creation,
ANSI-coloured log fragments, run-directory timestamp, file removal) -- humans grep these logs."""
import datetime
import os

_COLOURS = ("black", "red", "green", "yellow", "blue", "pink", "cyan", "white")


def ensure_dir(dir_path):
    os.makedirs(dir_path, exist_ok=True)


def set_color(log, color, highlight=True):
    """Wrap `log` in an ANSI colour escape; unknown colour names fall back to white (utils.py:10-22)."""
    code = _COLOURS.index(color) if color in _COLOURS else len(_COLOURS) - 1
    return "\033[" + ("1;3" if highlight else "0;3") + str(code) + "m" + log + "\033[0m"


def get_local_time():
    """Run-directory name, e.g. Oct-04-2026_05-31-07 (utils.py:24-33)."""
    return datetime.datetime.now().strftime("%b-%d-%Y_%H-%M-%S")


def delete_file(filename):
    if os.path.exists(filename):
        os.remove(filename)
