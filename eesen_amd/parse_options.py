"""The command-line conventions of the reference's tools for the Python hosts (train_ctc_parallel.py, net_output_extract.py).

Behaviour of /root/reference/src/util/parse-options.{h,cc}, the same as csrc/tools/parse_options.h implements for the native tools:
`--name=value` (a bare `--name` for booleans only), names case-insensitive with `_` == `-`, named options BEFORE the positional
arguments (a lone `--` ends them), the standard options --config=<file> (one `--x=y` per line, `#` comments; read first, so the
command line wins), --print-args (default true: echo of the command line on stderr), --help (usage, exit 0), --verbose; integers in
base 0; anything else prints the usage with the command line and raises ParseError("Invalid option ...") -- the tools print it and
return 255 (the reference's -1, train-ctc-parallel.cc:259-263).
"""
from __future__ import annotations

import sys
from types import SimpleNamespace
from typing import List, Optional


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
            try:
                o[2] = int(value, 0)
            except ValueError:
                self._bad(f'Invalid integer option "{value}"')
        elif kind in ("float", "double"):
            try:
                o[2] = float(value)
            except ValueError:
                self._bad(f'Invalid floating-point option "{value}"')
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
