"""The command-line conventions of the reference's tools for the Python hosts (train_ctc_parallel.py, net_output_extract.py).

Behaviour of /root/reference/src/util/parse-options.{h,cc}, the same as csrc/tools/parse_options.h implements for the native tools:
`--name=value` (a bare `--name` for booleans only), names case-insensitive with `_` == `-`, named options BEFORE the positional
arguments (a lone `--` ends them), the standard options --config=<file> (one `--x=y` per line, `#` comments; read first, so the
command line wins), --print-args (default true: echo of the command line on stderr), --help (usage, exit 0), --verbose; integers in
base 0; anything else prints the usage with the command line and raises ParseError("Invalid option ...") -- the tools print it and
return 255 (the reference's -1, train-ctc-parallel.cc:259-263).
"""
from __future__ import annotations

import re
import sys
from types import SimpleNamespace
from typing import List, Optional


# The reference converts option values with strtol(base 0) / strtod and only checks that SOME prefix parsed (parse-options.cc:561-
# 656; csrc/tools/parse_options.h does the same): leading white space, the longest valid numeric prefix, trailing garbage ignored,
# a leading 0 means octal.  int() / float() are stricter in some places ("010", "12abc", "1.5x") and laxer in others ("1_0"), so the
# Python trainer parses the way the C library does.
_INT_RE = re.compile(r"\s*([+-]?)(0[xX][0-9a-fA-F]+|0[0-7]*|[1-9][0-9]*)")
_FLT_RE = re.compile(r"\s*([+-]?(?:0[xX](?:[0-9a-fA-F]+\.?[0-9a-fA-F]*|\.[0-9a-fA-F]+)(?:[pP][+-]?[0-9]+)?|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?|inf(?:inity)?|nan(?:\([0-9a-zA-Z_]*\))?))", re.I)


def _strtol0(text: str):
    m = _INT_RE.match(text)
    if not m:
        return None
    sign, digits = m.group(1), m.group(2)
    if digits[:2].lower() == "0x":
        v = int(digits, 16)
    elif digits.startswith("0"):
        v = int(digits, 8) if len(digits) > 1 else 0
    else:
        v = int(digits)
    v = -v if sign == "-" else v
    v = max(-(1 << 63), min((1 << 63) - 1, v))                 # strtol saturates at LONG_MIN / LONG_MAX ...
    return ((v + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)          # ... and the (int32) cast of the reference wraps


def _strtod(text: str):
    m = _FLT_RE.match(text)
    if not m:
        return None
    tok = m.group(1)
    low = tok.lower().lstrip("+-")
    try:
        if low.startswith("0x"):
            return float.fromhex(tok)
        if low.startswith("nan"):
            return float("nan")
        return float(tok)
    except (ValueError, OverflowError):
        return float("inf") if not tok.startswith("-") else float("-inf")


class ParseError(RuntimeError):
    pass


def _norm(s: str) -> str:
    return s.replace("_", "-").lower()


def _escape(s: str) -> str:
    ok = "[]~#^_-+=:.,/"
    if s and all(c.isalnum() or c in ok for c in s):
        return s
    q = "'" if "'" not in s else '"'
    body = "".join(("\\" + c) if (q == '"' and c in '"\\$`') else c for c in s)
    return q + body + q


class ParseOptions:
    def __init__(self, usage: str, prog: str = ""):
        self.usage, self.prog = usage, prog
        self._opts = {}      # normalised name -> [name, kind, value, doc, standard]
        self.args: List[str] = []
        self._argv: List[str] = []
        self.register("config", "", "Configuration file to read (this option may be repeated)", standard=True)
        self.register("print-args", True, "Print the command line arguments (to stderr)", standard=True)
        self.register("help", False, "Print out usage message", standard=True)
        self.register("verbose", 0, "Verbose level (higher->more logging)", standard=True)

    def register(self, name: str, default, doc: str, kind: Optional[str] = None, standard: bool = False):
        """kind: bool | int | float | double | string (inferred from the default when omitted: a Python float is `float`)."""
        if kind is None:
            kind = "bool" if isinstance(default, bool) else "int" if isinstance(default, int) else "float" if isinstance(default, float) else "string"
        if kind == "bool":
            shown = "true" if default else "false"
        elif kind == "string":
            shown = '"%s"' % default
        else:
            shown = ("%g" % default)
        self._opts[_norm(name)] = [name, kind, default, f"{doc} ({kind}, default = {shown})", standard]

    # ------------------------------------------------------------------------------------------------------------------
    def print_usage(self, command_line: bool = False, file=None):
        f = file or sys.stderr
        print("\n" + self.usage, file=f)
        app = [(v[0], v[3]) for k, v in sorted(self._opts.items()) if not v[4]]
        if app:
            print("Options:", file=f)
            for n, d in app:
                print(f"  --{n:<25} : {d}", file=f)
            print("", file=f)
        print("Standard options:", file=f)
        for k, v in sorted(self._opts.items()):
            if v[4]:
                print(f"  --{v[0]:<25} : {v[3]}", file=f)
        print("", file=f)
        if command_line:
            print("Command line was: " + " ".join(_escape(a) for a in self._argv) + " ", file=f)
        f.flush()

    def _bad(self, msg: str):
        self.print_usage(True)
        raise ParseError(msg)

    def _split(self, arg: str):
        pos = arg.find("=")
        if pos < 0:
            return _norm(arg[2:]), "", False
        if pos == 2:
            self._bad("Invalid option (no key): " + arg)
        return _norm(arg[2:pos]), arg[pos + 1:].strip(), True

    def _set(self, key: str, value: str, eq: bool) -> bool:
        o = self._opts.get(key)
        if o is None:
            return False
        kind = o[1]
        if kind == "bool":
            if eq and value == "":
                raise ParseError(f"Invalid option --{key}=")
            v = value.lower()
            if v in ("true", "t", "1", ""):
                o[2] = True
            elif v in ("false", "f", "0"):
                o[2] = False
            else:
                self._bad("Invalid format for boolean argument [expected true or false]: " + value)
        elif kind == "int":
            v = _strtol0(value)
            if v is None:
                self._bad(f'Invalid integer option "{value}"')
            o[2] = v
        elif kind in ("float", "double"):
            v = _strtod(value)
            if v is None:
                self._bad(f'Invalid floating-point option "{value}"')
            o[2] = v
        else:
            if not eq:
                raise ParseError(f"Invalid option --{key}")
            o[2] = value
        return True

    def _read_config(self, filename: str):
        try:
            lines = open(filename).read().splitlines()
        except OSError:
            raise ParseError("Cannot open config file: " + filename)
        for n, line in enumerate(lines, 1):
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            if not line.startswith("--"):
                raise ParseError(f"Reading config file {filename}: line {n} does not look like a line from a command-line program's "
                                 "config file: should be of the form --x=y.  Note: config files intended to be sourced by shell scripts lack the '--'.")
            k, v, eq = self._split(line)
            if not self._set(k, v, eq):
                self._bad(f"Invalid option {line} in config file {filename}")

    def read(self, argv: Optional[List[str]] = None) -> SimpleNamespace:
        """argv WITHOUT the program name (sys.argv[1:] by default).  Returns a namespace: option names with `-` -> `_`, plus .args."""
        rest = list(sys.argv[1:] if argv is None else argv)
        self._argv = [self.prog or sys.argv[0]] + rest
        for a in rest:                               # first pass: --config, --help
            if not a.startswith("--"):
                continue
            if a == "--":
                break
            k, v, eq = self._split(a)
            if k == "config":
                self._read_config(v)
            if k == "help":
                self.print_usage()
                sys.exit(0)
        i, dd = 0, False
        while i < len(rest):                         # second pass: named options up to the first positional argument
            a = rest[i]
            if not a.startswith("--"):
                break
            if a == "--":
                i += 1; dd = True
                break
            k, v, eq = self._split(a)
            if not self._set(k, v, eq):
                self._bad("Invalid option " + a)
            i += 1
        for a in rest[i:]:
            if a == "--" and not dd:
                dd = True
            else:
                self.args.append(a)
        if self._opts["print-args"][2]:
            print(" ".join(_escape(a) for a in self._argv) + " ", file=sys.stderr, flush=True)
        ns = SimpleNamespace(**{v[0].replace("-", "_"): v[2] for v in self._opts.values()})
        ns.args = list(self.args)
        return ns
