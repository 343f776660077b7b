"""rocprofv3 (ROCm 7.2) leaves kernel symbols with `_Float16` parameters mangled (Itanium `DF16_`): a minimal demangler for this library's
own kernels - namespace, name and integer / bool template arguments, which is all the tables key on:
    _ZN3mvs17gl_entropy_kernelILi0ELi8ELi8ELb0ELb1ELb1EEEvPKv...  ->  void mvs::gl_entropy_kernel<0, 8, 8, false, true, true>(...)"""
import re


def demangle_mvs(name: str) -> str:
    m = re.match(r"_ZN3mvs(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    i = m.end()
    ident, rest = name[i:i + n], name[i + n:]
    if not rest.startswith("I"):
        return "mvs::%s(...)" % ident if rest.startswith("E") else name
    args, j = [], 1
    while j < len(rest) and rest[j] != "E":
        lit = re.match(r"L([ibjlm])(n?\d+)E", rest[j:])
        if not lit:
            return name                                    # a type argument: leave the symbol as it is
        v = lit.group(2).replace("n", "-")
        args.append(("true" if v != "0" else "false") if lit.group(1) == "b" else v)
        j += lit.end()
    return "void mvs::%s<%s>(...)" % (ident, ", ".join(args))


if __name__ == "__main__":
    import sys
    for a in sys.argv[1:]:
        print(demangle_mvs(a))
