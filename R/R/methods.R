# S3 surface of a "bess" object (the reference's R/R/coef.bess.R, predict.bess.R, print.bess.R, summary.bess.R,
# deviance.bess.R, logLik.bess.R): post-processing of the list bess() returns, untouched by which solver produced
# it.  Restated compactly from reading the reference; not executed in the build image (no R there).
# plot.bess (542 lines of base-graphics layout, R/R/plot.bess.R) is not restated: it draws the same fields
# (loss.all / ic.all / beta.all against s.list or lambda.list) and works on these objects unchanged if taken from the
# reference package.

coef.bess <- function(object, sparse = TRUE, ...) {
  beta <- if (!is.null(object$coef0)) c("(intercept)" = object$coef0, object$beta) else object$beta
  if (!sparse) return(beta)
  # the reference returns a one-column sparse Matrix (R/R/coef.bess.R:42-46); fall back to a plain matrix without it
  m <- matrix(beta, ncol = 1, dimnames = list(names(beta), NULL))
  if (requireNamespace("Matrix", quietly = TRUE)) Matrix::Matrix(m, sparse = TRUE) else m
}

predict.bess <- function(object, newx, type = c("link", "response"), ...) {
  if (missing(newx)) newx <- object$x
  if (is.null(colnames(newx))) {
    newx <- as.matrix(newx)
  } else {
    vn <- names(object$beta)
    if (any(is.na(match(vn, colnames(newx))))) stop("names of newx don't match training data!")
    newx <- as.matrix(newx[, vn])
  }
  type <- match.arg(type)
  eta <- newx %*% object$beta
  switch(object$family,
    gaussian = drop(eta) + object$coef0,                       # fitted values for either type
    binomial = {
      eta <- eta + object$coef0
      if (type == "link") drop(eta)
      else { e <- exp(eta); drop(ifelse(is.infinite(e), 1, e / (1 + e))) }
    },
    # the reference adds no intercept for these two (R/R/predict.bess.R:142-166)
    poisson = if (type == "link") eta else drop(exp(eta)),
    cox = if (type == "link") eta else drop(exp(eta)))
}

deviance.bess <- function(object, best.model = TRUE, ...) {
  n <- object$nsample
  gaussian <- object$family == "gaussian"
  if (best.model) {
    d <- if (gaussian) n * log(object$loss / 2) else object$loss
    names(d) <- "deviance"
    return(d)
  }
  if (!is.null(object$bess.one)) stop("Please set best.model = TRUE for bess objects from bess.one function.")
  loss <- if (object$method == "sequential") matrix(unlist(object$loss.all), nrow = length(object$s.list))
          else as.vector(unlist(object$loss.all))
  if (gaussian) n * loss else loss  # (sic: no log on the path values, R/R/deviance.bess.R:99-110)
}

logLik.bess <- function(object, best.model = TRUE, ...) {
  n <- object$nsample
  gaussian <- object$family == "gaussian"
  if (best.model) {
    ll <- if (gaussian) -n / 2 * (log(2 * pi) + log(object$loss) + 1) else -object$loss / 2
    names(ll) <- "Loglik"
    class(ll) <- "logLik"
    return(ll)
  }
  if (!is.null(object$bess.one)) stop("Please set best.model = TRUE for bess objects from bess.one function.")
  d <- deviance(object, best.model = FALSE)
  if (gaussian) -n / 2 * (log(2 * pi) + log(exp(d / n) * 2) + 1) else -d / 2
}

print.bess <- function(x, digits = max(5, getOption("digits") - 5), nonzero = FALSE, ...) {
  cat("Call:\n", paste(deparse(x$call), sep = "\n", collapse = "\n"), "\n\n", sep = "")
  if (nonzero) {
    b <- coef(x, sparse = FALSE)
    print(round(b[b != 0], digits), ...)
  } else {
    print(round(coef(x), digits), ...)
  }
  cat("\n")
  invisible(x)
}

summary.bess <- function(object, ...) {
  bar <- strrep("-", if (is.null(object$bess.one)) 91 else 82)
  sel <- names(which(object$beta != 0))
  ridge <- object$algorithm_type %in% c("L0L2", "GL0L2")
  num <- function(label, v) cat("    ", label, if (v >= 0) " " else "", v, "\n", sep = "")
  cat(bar, "\n", sep = "")
  if (is.null(object$bess.one)) {
    how <- function(m) if (m == "gsection") "golden section" else "sequential"
    if (!ridge) {
      cat("    Primal-dual active algorithm with tuning parameter determined by", how(object$method), "method", "\n\n")
    } else if (object$method == "sequential") {
      cat("    Penalized Primal-dual active algorithm", "\n")
      cat("    with tuning parameter determined by", how(object$method), "method", "\n\n")
    } else {
      cat("    Penalized Primal-dual active algorithm with tuning parameter determined by", "\n")
      cat("    powell method using", how(object$line.search), "method for line search", "\n\n")
    }
  }
  if (object$algorithm_type == "PDAS") cat("    Best model with k =", length(sel), "includes predictors:", "\n\n")
  else cat("    Best model with k =", length(sel), "lambda =", object$lambda, "includes predictors:", "\n\n")
  print(object$beta[sel])
  cat("\n")
  num("log-likelihood:   ", logLik(object))
  num("deviance:         ", deviance(object))
  if (is.null(object$bess.one)) {
    if (object$ic.type == "cv") num("cv loss:          ", object$cvm)
    else num(sprintf("%-18s", paste0(object$ic.type, ":")), object$ic)
  }
  cat(bar, "\n", sep = "")
  invisible(object)
}
