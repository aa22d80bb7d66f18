"""The R package's files cannot be executed here (no R, no Rcpp): what can be checked is that they fit together --
the argument list of bessCpp() in the shim, in the hand-written RcppExports glue and in the R stub are the same 30
names in the same order as the reference's bessCpp (src/bess.h:20-33), the registration table names the entry point
NAMESPACE's useDynLib(.registration = TRUE) will look up, and every function NAMESPACE exports is defined."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ARGS = ["x", "y", "data_type", "weight", "is_normal", "algorithm_type", "model_type", "max_iter", "exchange_num",
            "path_type", "is_warm_start", "ic_type", "is_cv", "K", "state", "sequence", "lambda_seq", "s_min", "s_max",
            "K_max", "epsilon", "lambda_min", "lambda_max", "nlambda", "is_screening", "screening_size", "powell_path",
            "g_index", "always_select", "tao"]


def _read(*parts):
    return open(os.path.join(ROOT, *parts)).read()


def _arg_names(decl):
    return [a.strip().split()[-1] for a in decl.split(",")]


def test_bessCpp_argument_lists_agree():
    shim = _read("R", "src", "bess_amd_shim.cpp")
    glue = _read("R", "src", "RcppExports.cpp")
    stub = _read("R", "R", "RcppExports.R")
    m = re.search(r"// \[\[Rcpp::export\]\]\s*Rcpp::List bessCpp\((.*?)\)\s*\{", shim, re.S)
    assert m and _arg_names(m.group(1)) == REF_ARGS
    m = re.search(r"Rcpp::List bessCpp\((.*?)\);", glue, re.S)
    assert m and _arg_names(m.group(1)) == REF_ARGS
    m = re.search(r"RcppExport SEXP _BeSSamd_bessCpp\((.*?)\)\s*\{", glue, re.S)
    assert m and [a.replace("SEXP", "").strip() for a in m.group(1).split(",")] == REF_ARGS
    assert re.search(r'\{"_BeSSamd_bessCpp", \(DL_FUNC\) &_BeSSamd_bessCpp, 30\}', glue)
    assert "R_init_BeSSamd" in glue and "R_registerRoutines" in glue
    m = re.search(r"bessCpp <- function\((.*?)\)\s*\{\s*\.Call\(`_BeSSamd_bessCpp`,(.*?)\)\s*\}", stub, re.S)
    assert m and _arg_names(m.group(1)) == REF_ARGS and _arg_names(m.group(2)) == REF_ARGS


def test_namespace_exports_are_defined():
    ns = _read("R", "NAMESPACE")
    code = "\n".join(_read("R", "R", f) for f in sorted(os.listdir(os.path.join(ROOT, "R", "R"))))
    assert "useDynLib(BeSSamd, .registration = TRUE)" in ns
    for name in re.search(r"export\((.*?)\)", ns).group(1).split(","):
        assert re.search(r"^%s\s*<-\s*function" % re.escape(name.strip()), code, re.M), name
    for gen, cls in re.findall(r"S3method\((\w+),\s*(\w+)\)", ns):
        assert re.search(r"^%s\.%s\s*<-\s*function" % (gen, cls), code, re.M), (gen, cls)
