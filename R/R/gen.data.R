# gen.data(): the reference's simulator (R/R/gen.data.R:93-255), restated.  Coefficient magnitudes follow the same
# recipe as bess_amd/synth.py (python/bess/gen_data.py): m = 5 sqrt(2 log p / n); gaussian U(m, 100 m), the other
# families U(2m, 10m).  cortype 1: AR(1) correlation rho^|i-j|; 2: exchangeable rho; 3: the banded moving-average
# design X + rho (X shifted left + X shifted right) on standardised columns.  Not executed in the build image.
gen.data <- function(n, p, k = NULL, rho = 0, family = c("gaussian", "binomial", "poisson", "cox"),
                     beta = NULL, cortype = 1, snr = 10, censoring = TRUE, c = 1, scal, sigma = 1, seed = 1) {
  family <- match.arg(family)
  set.seed(seed)
  on.exit(set.seed(NULL), add = TRUE)
  if (!is.null(beta)) k <- sum(abs(beta) > 1e-5) else if (is.null(k)) stop("Please provide an integer to k.")
  survival_times <- function(eta) {
    time <- (-log(runif(n)) / drop(exp(eta)))^(1 / scal)
    if (censoring) {
      ctime <- c * runif(n)
      status <- (time < ctime) * 1
      cat("censoring rate:", 1 - sum(status) / n, "\n")
      time <- pmin(time, ctime)
    } else {
      status <- rep(1, times = n)
      cat("no censoring", "\n")
    }
    cbind(time = time, status = status)
  }
  clip30 <- function(v) pmin(pmax(v, -30), 30)
  if (cortype != 3) {
    Sigma <- if (cortype == 1) rho^abs(outer(1:p, 1:p, "-")) else matrix(rho, p, p) + diag(1 - rho, p, p)
    x <- MASS::mvrnorm(n, rep(0, p), Sigma)
    nonzero <- sample(1:p, k)
    Tbeta <- rep(0, p)
    m <- 5 * sqrt(2 * log(p) / n)
    draw <- function(lo, hi) if (is.null(beta)) { Tbeta[nonzero] <<- runif(k, lo, hi) } else { Tbeta <<- beta }
    noise_sd <- function() sqrt(drop(t(Tbeta) %*% Sigma %*% Tbeta) / snr)
    if (family == "gaussian") {
      draw(m, 100 * m)
      y <- x %*% Tbeta + rnorm(n, 0, noise_sd())
    } else if (family == "binomial") {
      draw(2 * m, 10 * m)
      eta <- x %*% Tbeta + rnorm(n, 0, noise_sd())
      pr <- ifelse(is.infinite(exp(eta)), 1, exp(eta) / (1 + exp(eta)))
      y <- rbinom(n, 1, pr)
    } else if (family == "cox") {
      draw(2 * m, 10 * m)
      y <- survival_times(x %*% Tbeta)
    } else {
      x <- x / 16
      m <- 5 * sigma * sqrt(2 * log(p) / n)
      draw(2 * m, 10 * m)
      y <- rpois(n, exp(clip30(x %*% Tbeta + rnorm(n, 0, noise_sd()))))
    }
    return(list(x = x, y = y, Tbeta = Tbeta))
  }
  X <- scale(matrix(rnorm(n * p), n, p), TRUE, FALSE)
  X <- sqrt(n) * scale(X, FALSE, sqrt(colSums(X^2)))
  zero <- rep(0, n)
  x <- X + rho * (cbind(zero, X[, 1:(p - 2)], zero) + cbind(zero, X[, 3:p], zero))
  colnames(x) <- paste0("X", 1:ncol(x))
  nonzero <- sample(1:p, k)
  Tbeta <- rep(0, p)
  m <- 5 * (if (family == "gaussian") 1 else sigma) * sqrt(2 * log(p) / n)
  if (is.null(beta)) Tbeta[nonzero] <- if (family == "gaussian") runif(k, m, 100 * m) else runif(k, 2 * m, 10 * m)
  else Tbeta <- beta
  eta <- drop(x %*% Tbeta)
  y <- switch(family,
    gaussian = eta + rnorm(n, 0, sigma^2),
    binomial = rbinom(n = n, size = 1, prob = exp(eta) / (1 + exp(eta))),
    cox = survival_times(eta),
    poisson = rpois(n, exp(clip30(eta))))
  list(x = x, y = y, Tbeta = Tbeta)
}
