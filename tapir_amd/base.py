"""Command-line plumbing of tapir_compute.py: argparse types and actions, output-directory naming, file discovery.

The names and the observable behaviour (return values, exception types) are those of the reference's helper module,
because the command line is the drop-in surface: FullPaths (tapir/base.py:19-22), is_dir
(:27-33), create_unique_dir (:41-66), get_output_type (:68-73), get_list_from_ints (:75-82),
get_strings_from_items (:84-91), get_list_from_ranges (:93-100), get_files (:102-114),
parse_subset_map_file (:116-123).  The implementations are this project's own.
"""
import argparse
import csv
import glob
import itertools
import os
from pathlib import Path

_IMAGE_TYPES = frozenset(("pdf", "png", "tiff", "jpeg", "jpg"))


def _absolute(path):
    return os.path.abspath(os.path.expanduser(path))


class FullPaths(argparse.Action):
    """argparse action: store the argument as an absolute path, `~` expanded."""

    def __call__(self, parser, namespace, values, option_string=None):
        setattr(namespace, self.dest, _absolute(values))


def is_dir(dirname):
    """argparse type: an existing directory, returned unchanged; anything else is an ArgumentTypeError."""
    if Path(dirname).is_dir():
        return dirname
    raise argparse.ArgumentTypeError("{0} is not a directory".format(dirname))


def create_unique_dir(path, limit=100):
    """Directory for this run's outputs.  `path` must exist; when it is empty it is used as it is, otherwise the first
    of `path.1`, `path.2`, ... that can be created is, giving up after `limit` candidates."""
    if not any(Path(path).iterdir()):
        return path
    for n in itertools.islice(itertools.count(1), max(0, limit - 1)):
        candidate = "{0}.{1}".format(path, n)
        try:
            os.mkdir(candidate)
        except FileExistsError:
            continue
        return candidate
    raise Exception("could not uniquely create directory {0}: limit `{1}` reached".format(path, limit))


def get_output_type(name):
    """Image type of an output file name, from its extension (AssertionError when it is not a supported one)."""
    ext = Path(name).suffix[1:].lower()
    if ext not in _IMAGE_TYPES:
        raise AssertionError("Filetype must be one of pdf, png, tiff, or jpeg")
    return ext


def _comma_separated(string, convert, what, name):
    out = []
    for token in string.split(","):
        try:
            out.append(convert(token))
        except (TypeError, ValueError) as err:
            raise argparse.ArgumentTypeError("Cannot convert {0} to a list of {1}: {2}".format(name, what, err))
    return out


def get_list_from_ints(string, name="time"):
    """'10,20,50' -> [10, 20, 50] (the --times option)."""
    return _comma_separated(string, int, "integers", name)


def get_strings_from_items(string, name="locus"):
    """'a,b,c' -> ['a', 'b', 'c']."""
    return _comma_separated(string, str, "loci", name)


def get_list_from_ranges(string):
    """'0-10,20-100' -> [[0, 10], [20, 100]] (the --intervals option)."""
    def span(token):
        return [int(bound) for bound in token.split("-")]
    return _comma_separated(string, span, "integers", "spans")


def get_files(d, extension):
    """Files of directory `d` matching one glob pattern or several separated by commas, in glob order per pattern (the
    order the loci are processed and stored in); IOError when nothing matches."""
    patterns = [p for p in extension.strip(" ").split(",")]
    found = list(itertools.chain.from_iterable(glob.iglob(os.path.join(d, p)) for p in patterns))
    if not found:
        raise IOError("There appear to be no files of {0} type in {1}".format(patterns, d))
    return found


def parse_subset_map_file(filename):
    """--subset-pi-map-file: tab-separated `alignment name, start, end` (0-offset, end exclusive) per line; yields
    (name, [start, end]); blank lines are skipped."""
    with open(filename, newline="") as handle:
        for row in csv.reader(handle, delimiter="\t"):
            if not row or not "".join(row).strip():
                continue
            name, start, end = (field.strip() for field in row[:3])
            yield name, [int(start), int(end)]
